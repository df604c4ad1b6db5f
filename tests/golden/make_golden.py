#!/usr/bin/env python3
"""
Generates the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in the dev container.

  python tests/golden/make_golden.py            # needs /root/reference; never runs on the GPU box

What is imported from /root/reference, and how:
  * ghn3/graphormer.py  -- imported UNMODIFIED by file path (torch-only module).  Produces
                           ``graphormer_*.npz``: a true oracle for SURVEY rows A4-A8.
  * ghn3 (package)      -- imported with stand-in modules for the third-party packages that are not
                           installed in this image (``ppuda``, ``torchvision``, ``h5py``); the ppuda
                           base classes (GHN, ConvDecoder, ShapeEncoder, MLP, named_layered_modules)
                           are the restatement in oracle/ppuda_base.py.  The reference's own
                           ``GHN3.forward / _map_net_params / _tile_params / _normalize /
                           _set_params / ConvDecoder3.forward / GraphBatch`` then run as shipped.
                           Produces ``ghn3_tiny_*.npz`` and ``tile_cases.npz``.

Only data (inputs, expected outputs) is written; no reference source text is stored.
GHN weights are NOT stored: both this script and the tests rebuild them with
``tests/golden/recipe.py::seeded_state_dict`` (numpy RandomState, stable bit stream).
"""

import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import ppuda_base                              # noqa: E402
import recipe                                              # noqa: E402


# ------------------------------------------------------------------------------------------------
# stand-ins for packages absent from the image
# ------------------------------------------------------------------------------------------------

def install_standins():
    import transformers, transformers.pytorch_utils        # noqa: F401  must precede the torchvision stub

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Stub(nn.Module):
        pass

    names = ['Inception3', 'SwinTransformer', 'VisionTransformer', 'SqueezeNet']
    vt = mod('torchvision.models.vision_transformer', Encoder=type('Encoder', (nn.Module,), {}))
    cn = mod('torchvision.models.convnext', LayerNorm2d=type('LayerNorm2d', (nn.Module,), {}))
    tvm = mod('torchvision.models', vision_transformer=vt, convnext=cn,
              **{n: type(n, (nn.Module,), {}) for n in names})
    tvt = mod('torchvision.transforms')
    mod('torchvision', models=tvm, transforms=tvt)
    mod('h5py')

    mod('ppuda')
    mod('ppuda.ghn')
    mod('ppuda.ghn.nn', GHN=ppuda_base.GHN, ConvDecoder=ppuda_base.ConvDecoder)

    class AvgrageMeter:
        pass

    mod('ppuda.utils', capacity=ppuda_base.capacity, AvgrageMeter=AvgrageMeter, accuracy=None, init=None,
        rand_choice=None, infer=None, adjust_net=None)
    mod('ppuda.deepnets1m')
    mod('ppuda.deepnets1m.net', named_layered_modules=ppuda_base.named_layered_modules,
        get_cell_ind=lambda *a, **k: 0, Network=type('Network', (nn.Module,), {}),
        AuxiliaryHeadImageNet=ppuda_base.AuxiliaryHeadImageNet, AuxiliaryHeadCIFAR=ppuda_base.AuxiliaryHeadCIFAR,
        drop_path=ppuda_base.drop_path, _is_none=ppuda_base.is_none)
    mod('ppuda.deepnets1m.ops', parse_op_ks=ppuda_base.parse_op_ks, PosEnc=type('PosEnc', (nn.Module,), {}))
    mod('ppuda.deepnets1m.genotypes', PRIMITIVES_DEEPNETS1M=ppuda_base.PRIMITIVES_DEEPNETS1M, from_dict=None)
    mod('ppuda.deepnets1m.loader', DeepNets1M=object, NetBatchSampler=object, MAX_NODES_BATCH=2200)


def load_reference_graphormer():
    spec = importlib.util.spec_from_file_location('_ref_graphormer', os.path.join(REF, 'ghn3', 'graphormer.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.create_transformer(nn.Module, nn.Linear, nn.GELU, nn.ReLU, nn.LayerNorm, nn.Dropout, nn.Identity,
                                nn.Sequential)


# ------------------------------------------------------------------------------------------------
# 1. Graphormer goldens (reference graphormer.py unmodified)
# ------------------------------------------------------------------------------------------------

def make_graphormer_goldens():
    types_ = load_reference_graphormer()
    Layer = types_['TransformerLayer']
    out = {}
    for tag, (C, H, N, B) in {'c32': (32, 4, 12, 2), 'c48': (48, 16, 70, 1)}.items():
        rs = np.random.RandomState(1234 + C)
        l0 = Layer(dim=C, edge_dim=2, num_heads=H, mlp_ratio=4, return_edges=True)
        l0.centrality_embed_in = nn.Embedding(101, C)
        l0.centrality_embed_out = nn.Embedding(101, C)
        l0.input_dist_embed = nn.Embedding(1001, C)
        l1 = Layer(dim=C, edge_dim=0, num_heads=H, mlp_ratio=4, return_edges=False)
        for li, layer in enumerate((l0, l1)):
            sd = recipe.seeded_state_dict({k: tuple(v.shape) for k, v in layer.state_dict().items()},
                                          seed=100 * C + li)
            layer.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        n_nodes = [N, max(3, N - 5)][:B]
        A = np.zeros((B, N, N), dtype=np.int64)
        for b in range(B):
            A[b, :n_nodes[b], :n_nodes[b]] = recipe.random_dag_spd(n_nodes[b], seed=77 + b + C, cutoff=50)
        x = rs.standard_normal((B, N, C)).astype(np.float32)
        node_mask = np.zeros((B, N, 1), dtype=bool)
        for b in range(B):
            node_mask[b, :n_nodes[b]] = True
        mask = torch.from_numpy(node_mask) & torch.from_numpy(node_mask).permute(0, 2, 1)
        with torch.no_grad():
            y0, bias, _ = l0(torch.from_numpy(x.copy()), torch.from_numpy(A), mask)
            y1 = l1(y0.clone(), bias, mask)
        out[tag + '/x'] = x
        out[tag + '/A'] = A.astype(np.int16)
        out[tag + '/n_nodes'] = np.asarray(n_nodes, dtype=np.int64)
        out[tag + '/cfg'] = np.asarray([C, H, N, B], dtype=np.int64)
        out[tag + '/y0'] = y0.numpy()
        out[tag + '/bias'] = bias.numpy()
        out[tag + '/y1'] = y1.numpy()
    np.savez_compressed(os.path.join(HERE, 'graphormer_layers.npz'), **out)
    print('graphormer_layers.npz', {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------------------------------------
# 2. Full GHN-3 forward (+ gradient summaries) on the tiny config, reference GHN3 class
# ------------------------------------------------------------------------------------------------

def make_ghn3_goldens():
    install_standins()
    sys.path.insert(0, REF)
    import ghn3                                              # noqa: F401  the reference package
    from ghn3.nn import GHN3
    from ghn3.graph import Graph, GraphBatch
    Encoder = sys.modules['torchvision.models.vision_transformer'].Encoder

    cfg = recipe.TINY_CFG
    torch.manual_seed(0)
    ghn = GHN3(**cfg, debug_level=0)
    shapes = {k: tuple(v.shape) for k, v in ghn.state_dict().items()}
    sd = recipe.seeded_state_dict(shapes, seed=recipe.TINY_SEED)
    ghn.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    n_params = sum(p.numel() for p in ghn.parameters())

    out = {'meta/n_params': np.asarray([n_params], dtype=np.int64)}
    names_sorted = sorted(shapes)
    out['meta/state_keys'] = np.asarray(names_sorted)
    out['meta/state_shapes'] = np.asarray([str(shapes[k]) for k in names_sorted])

    for case, spec_ids in recipe.TINY_CASES.items():
        specs = [recipe.TINY_NETS[i] for i in spec_ids]
        nets = [recipe.build_torch_net(s, encoder_cls=Encoder) for s in specs]
        graphs = []
        for s in specs:
            node_feat, node_info, A = recipe.graph_arrays(s)
            graphs.append(Graph(node_feat=torch.from_numpy(node_feat), node_info=node_info,
                                A=torch.from_numpy(A), dense=True))
        batch = GraphBatch(graphs, dense=True)
        ghn.train()
        ghn.zero_grad()
        torch.manual_seed(5)
        nets_out, emb = ghn(nets, batch, return_embeddings=True, keep_grads=True,
                            bn_track_running_stats=True, reduce_graph=False)
        loss = 0
        for b, net in enumerate(nets_out):
            for name, p in recipe.named_predicted(net):
                out['%s/pred/%d/%s' % (case, b, name)] = p.detach().numpy()
                q = p[:, 1:] if p.dim() == 3 else p            # Q3: skip the random class-token row
                loss = loss + torch.norm(q, p='fro')
        out[case + '/emb'] = emb.detach().numpy()
        out[case + '/loss'] = np.asarray([loss.item()], dtype=np.float64)
        loss.backward()
        for k, p in ghn.named_parameters():
            g = p.grad
            assert g is not None, k
            idx = recipe.sample_indices(g.numel(), 8, seed=len(k))
            out['%s/grad/%s' % (case, k)] = np.concatenate(
                [[g.norm().item(), g.sum().item()], g.reshape(-1)[idx].numpy()]).astype(np.float64)
    np.savez_compressed(os.path.join(HERE, 'ghn3_tiny.npz'), **out)
    print('ghn3_tiny.npz: %d arrays, GHN params %d' % (len(out), n_params))

    # ---- tile / normalise case table, straight from the reference methods --------------------
    tc = {}
    rs = np.random.RandomState(99)
    for i, (src, tgt) in enumerate(recipe.TILE_CASES):
        w = rs.standard_normal(src).astype(np.float32)
        torch.manual_seed(11)
        t = ghn._tile_params(torch.from_numpy(w), tgt)
        tc['%d/w' % i] = w
        tc['%d/tiled' % i] = t.numpy()
        for is_w in (0, 1):
            tc['%d/norm%d' % (i, is_w)] = ghn._normalize(nn.Identity(), t, bool(is_w)).numpy()
    np.savez_compressed(os.path.join(HERE, 'tile_cases.npz'), **tc)
    print('tile_cases.npz: %d arrays' % len(tc))

    # ---- structure pin: parameter counts of the four released variants ------------------------
    counts = {}
    for name, (hid, layers, heads) in recipe.VARIANTS.items():
        if hid > 128:
            # count without allocating: shapes follow from the tiny model's key set
            counts[name] = recipe.count_params(hid, layers, heads, 1000)
        else:
            m = GHN3(max_shape=(hid, hid, 16, 16), num_classes=1000, hid=hid, heads=heads, layers=layers,
                     weight_norm=True, ve=True, layernorm=True)
            counts[name] = sum(p.numel() for p in m.parameters())
            assert counts[name] == recipe.count_params(hid, layers, heads, 1000), name
    print('variant parameter counts:', counts)
    assert counts['ghn3xlm16'] == 654365184, counts     # examples/ghn_all_pytorch.ipynb:109


# ------------------------------------------------------------------------------------------------
# 2b. Extra tiny cases (recipe.EXTRA_CASES): kernels larger than the decoder grid (bilinear branch, nn.py:751-753)
#     and the weight_norm=False / layernorm=False configurations -> ghn3_tiny_extra.npz
# ------------------------------------------------------------------------------------------------

def make_extra_goldens():
    install_standins()
    sys.path.insert(0, REF)
    import ghn3                                              # noqa: F401  the reference package
    from ghn3.nn import GHN3
    from ghn3.graph import Graph, GraphBatch
    Encoder = sys.modules['torchvision.models.vision_transformer'].Encoder
    out = {}
    for case, (specs, over) in recipe.EXTRA_CASES.items():
        cfg = dict(recipe.TINY_CFG, **over)
        torch.manual_seed(0)
        ghn = GHN3(**cfg, debug_level=0)
        shapes = {k: tuple(v.shape) for k, v in ghn.state_dict().items()}
        sd = recipe.seeded_state_dict(shapes, seed=recipe.TINY_SEED)
        ghn.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        out[case + '/state_keys'] = np.asarray(sorted(shapes))
        nets = [recipe.build_torch_net(s, encoder_cls=Encoder) for s in specs]
        graphs = []
        for s in specs:
            node_feat, node_info, A = recipe.graph_arrays(s)
            graphs.append(Graph(node_feat=torch.from_numpy(node_feat), node_info=node_info,
                                A=torch.from_numpy(A), dense=True))
        batch = GraphBatch(graphs, dense=True)
        ghn.train()
        ghn.zero_grad()
        torch.manual_seed(5)
        nets_out, emb = ghn(nets, batch, return_embeddings=True, keep_grads=True,
                            bn_track_running_stats=True, reduce_graph=False)
        loss = 0
        for b, net in enumerate(nets_out):
            for name, p in recipe.named_predicted(net):
                out['%s/pred/%d/%s' % (case, b, name)] = p.detach().numpy()
                q = p[:, 1:] if p.dim() == 3 else p            # Q3: skip the random class-token row
                loss = loss + torch.norm(q, p='fro')
        out[case + '/emb'] = emb.detach().numpy()
        out[case + '/loss'] = np.asarray([loss.item()], dtype=np.float64)
        loss.backward()
        for k, p in ghn.named_parameters():
            g = p.grad
            assert g is not None, k
            idx = recipe.sample_indices(g.numel(), 8, seed=len(k))
            out['%s/grad/%s' % (case, k)] = np.concatenate(
                [[g.norm().item(), g.sum().item()], g.reshape(-1)[idx].numpy()]).astype(np.float64)
    np.savez_compressed(os.path.join(HERE, 'ghn3_tiny_extra.npz'), **out)
    print('ghn3_tiny_extra.npz: %d arrays' % len(out))


# ------------------------------------------------------------------------------------------------
# 2c. Automatic graph construction: the reference's Graph(model) (graph.py:292-908) on the hand-written networks of
#     tests/golden/graph_nets.py -> graphs.npz (node types, adjacency with virtual edges, node_info, shapes)
# ------------------------------------------------------------------------------------------------

def make_graph_goldens():
    install_standins()
    sys.path.insert(0, REF)
    import ghn3                                              # noqa: F401  the reference package
    from ghn3.graph import Graph
    import graph_nets
    tvm = sys.modules['torchvision.models']
    bases = {'VisionTransformer': tvm.VisionTransformer, 'Encoder': tvm.vision_transformer.Encoder}
    out = {}
    for name, net in graph_nets.all_nets(bases).items():
        for ve in (50, 1):
            g = Graph(net, ve_cutoff=ve, verbose=False)
            tag = '%s/ve%d' % (name, ve)
            out[tag + '/node_feat'] = g.node_feat.view(-1).numpy().astype(np.int16)
            out[tag + '/A'] = g._Adj.numpy().astype(np.int16)
            out[tag + '/names'] = np.asarray([n['param_name'] for n in g._nodes])
            out[tag + '/node_info'] = np.asarray([repr([(int(a), str(b), str(c), None if d is None else tuple(
                int(v) for v in d), bool(e), bool(f)) for (a, b, c, d, e, f) in cell]) for cell in g.node_info])
            out[tag + '/shapes'] = np.asarray([repr(None if s_ is None else tuple(int(v) for v in s_))
                                               for s_ in g._param_shapes])
            print(tag, 'nodes', g.n_nodes, 'edges', int((g._Adj == 1).sum()))
    np.savez_compressed(os.path.join(HERE, 'graphs.npz'), **out)
    print('graphs.npz: %d arrays' % len(out))


# ------------------------------------------------------------------------------------------------
# 2d. Target networks of the DeepNets-1M search space: the reference's Network (torch.nn flavour) and NetworkLight
#     (light layers) of ghn3/ops.py:306-585 on the genotypes of tests/golden/network_cases.py -> networks.npz:
#     seeded parameters (recipe.seeded_net_params), a seeded image batch, logits in training mode (batch statistics),
#     logits of the light flavour with the same tensors assigned, parameter names / shapes, gradient norms.
# ------------------------------------------------------------------------------------------------

def make_network_goldens():
    install_standins()
    sys.path.insert(0, REF)
    import ghn3                                              # noqa: F401
    from ghn3.ops import Network, NetworkLight
    from collections import namedtuple
    import network_cases
    Genotype = namedtuple('Genotype', 'normal normal_concat reduce reduce_concat')
    out = {}
    for name, (geno, kw, img) in network_cases.CASES.items():
        g = Genotype(**geno)
        torch.manual_seed(0)
        net = Network(genotype=g, **kw)
        names = [n for n, _ in net.named_parameters()]
        params = recipe.seeded_net_params([(n, tuple(p.shape)) for n, p in net.named_parameters()], seed=len(name))
        with torch.no_grad():
            for n, p in net.named_parameters():
                p.copy_(torch.from_numpy(params[n]))
        x = torch.from_numpy(recipe.seeded_images(img, seed=7))
        net.train()
        torch.manual_seed(123)                               # (Dropout in the two-layer classifier head)
        logits, aux = net(x)
        loss = logits.square().mean() + (aux.square().mean() if aux is not None else 0.)
        loss.backward()
        out[name + '/names'] = np.asarray(names)
        out[name + '/shapes'] = np.asarray([repr(tuple(p.shape)) for _, p in net.named_parameters()])
        out[name + '/logits'] = logits.detach().numpy()
        if aux is not None:
            out[name + '/aux'] = aux.detach().numpy()
        out[name + '/grad_norms'] = np.asarray([float(p.grad.norm()) if p.grad is not None else -1.0
                                                for _, p in net.named_parameters()], dtype=np.float64)
        # light flavour: same tensors assigned through the table GHN3.forward walks
        light = NetworkLight(genotype=g, **{k: ('bn' if (k == 'norm' and v) else v) for k, v in kw.items()})
        table = {}
        for cell in light._layered_modules:
            table.update(cell)
        out[name + '/light_names'] = np.asarray(sorted(table))
        out[name + '/light_cells'] = np.asarray([c for c, cell in enumerate(light._layered_modules) for _ in cell])
        for n, e in table.items():
            setattr(e['module'], 'weight' if e['is_w'] else 'bias', torch.from_numpy(params[n]))
        torch.manual_seed(123)
        ll, la = light(x)
        out[name + '/light_logits'] = ll.detach().numpy()
        print(name, 'params', sum(int(np.prod(p.shape)) for p in params.values()), 'logits', tuple(logits.shape),
              'light == torch:', float((ll - logits).detach().abs().max()))
    np.savez_compressed(os.path.join(HERE, 'networks.npz'), **out)
    print('networks.npz: %d arrays' % len(out))


# ------------------------------------------------------------------------------------------------
# 3. torchvision-shaped ResNets through the reference GHN3 class at the released sizes (BASELINE configs 1 and 4):
#    ghn3tm8 on the ResNet-18 graph, ghn3xlm16 on the ResNet-50 graph.  Stores per predicted tensor the Frobenius
#    norm and a seeded sample of its elements (the tensors themselves are 11.7 M / 25.6 M floats).
# ------------------------------------------------------------------------------------------------

def make_resnet_goldens(which=('18', '50', 'vit')):
    install_standins()
    sys.path.insert(0, REF)
    import ghn3                                              # noqa: F401  the reference package
    from ghn3.nn import GHN3
    from ghn3.graph import Graph, GraphBatch
    import time
    Encoder = sys.modules['torchvision.models.vision_transformer'].Encoder
    for depth, variant in ((18, 'ghn3tm8'), (50, 'ghn3xlm16'), ('vit', 'ghn3xlm16')):
        if str(depth) not in which:
            continue
        hid, layers, heads = recipe.VARIANTS[variant]
        cfg = dict(max_shape=(hid, hid, 16, 16), num_classes=1000, hid=hid, heads=heads, layers=layers,
                   weight_norm=True, ve=True, layernorm=True)
        torch.manual_seed(0)
        ghn = GHN3(**cfg, debug_level=0)
        shapes = {k: tuple(v.shape) for k, v in ghn.state_dict().items()}
        sd = recipe.seeded_state_dict(shapes, seed=recipe.RESNET_SEED)
        ghn.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        del sd
        spec = recipe.vit_b16_spec() if depth == 'vit' else recipe.resnet_spec(depth)
        net = recipe.build_torch_net(spec, encoder_cls=Encoder)
        node_feat, node_info, A = recipe.graph_arrays(spec)
        g = Graph(node_feat=torch.from_numpy(node_feat), node_info=node_info, A=torch.from_numpy(A), dense=True)
        batch = GraphBatch([g], dense=True)
        ghn.eval()
        t0 = time.time()
        with torch.no_grad():
            net_out, emb = ghn(net, batch, return_embeddings=True, bn_track_running_stats=True, reduce_graph=False)
        tag = 'vit_b16' if depth == 'vit' else 'resnet%d' % depth
        print('%s / %s: reference forward %.1f s' % (tag, variant, time.time() - t0))
        out = {'meta/variant': np.asarray([variant]), 'emb': emb.detach().numpy().astype(np.float32)}
        total = 0
        for name, p in recipe.named_predicted(net_out):
            v = p.detach().reshape(-1)
            total += v.numel()
            if v.dim() == 1 and p.dim() == 3:
                pass
            q = p.detach()[:, 1:].reshape(-1) if p.dim() == 3 else v       # Q3: row 0 of a positional encoding is random
            idx = recipe.sample_indices(q.numel(), recipe.RESNET_SAMPLES, seed=len(name))
            out['pred/%s/norm' % name] = np.asarray([float(q.double().norm())])
            out['pred/%s/sample' % name] = q[idx].numpy().astype(np.float32)
        out['meta/n_predicted'] = np.asarray([total], dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, '%s_%s.npz' % (tag, variant)), **out)
        print('%s_%s.npz: %d predicted params' % (tag, variant, total))


# ------------------------------------------------------------------------------------------------
# 4. The BENCHMARKED workload through the reference GHN3 class, forward AND backward: ghn3xlm16 on the seeded synthetic
#    256-node graph bench.py times (seed 256000) and on one ragged two-graph batch (quirk Q1: padded dense rows), loss =
#    sum of Frobenius norms of the predicted tensors (the reference's predparam_wd term, trainer.py:97-98,288-294).
#    Stores per predicted tensor the norm + a seeded sample, per GHN parameter the gradient norm + a seeded sample.
#    The graphs come from ghn3_amd.synthetic (inputs are data; the generator is numpy-only and seeded), the target networks
#    are its shape-only light networks, which the reference handles through `_layered_modules` (nn.py:612, 531-538).
# ------------------------------------------------------------------------------------------------

BENCH_CASES = {'b1': ('ghn3xlm16', [256], 256000), 'b2r': ('ghn3xlm16', [90, 170], 777000)}
BENCH_SEED = 31337
BENCH_SAMPLES = 2048


def make_bench_goldens(which=('b1', 'b2r')):
    install_standins()
    sys.path.insert(0, REF)
    import ghn3                                              # noqa: F401  the reference package
    from ghn3.nn import GHN3
    from ghn3.graph import Graph, GraphBatch
    from ghn3_amd.synthetic import synthetic_batch
    import time
    for case in which:
        variant, nodes, seed0 = BENCH_CASES[case]
        hid, layers, heads = recipe.VARIANTS[variant]
        cfg = dict(max_shape=(hid, hid, 16, 16), num_classes=1000, hid=hid, heads=heads, layers=layers,
                   weight_norm=True, ve=True, layernorm=True)
        torch.manual_seed(0)
        ghn = GHN3(**cfg, debug_level=0)
        shapes = {k: tuple(v.shape) for k, v in ghn.state_dict().items()}
        sd = recipe.seeded_state_dict(shapes, seed=BENCH_SEED)
        ghn.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        del sd
        gb, nets = synthetic_batch(nodes, seed0)
        graphs = [Graph(node_feat=g.node_feat.clone(), node_info=g.node_info, A=g._Adj.clone(), dense=True)
                  for g in gb.graphs]
        batch = GraphBatch(graphs, dense=True)
        ghn.train()
        t0 = time.time()
        ghn(nets, batch, keep_grads=True, bn_track_running_stats=True, reduce_graph=False)
        named = []
        for b, net in enumerate(nets):
            for lname, m in net.layers.items():
                for attr in ('weight', 'bias'):
                    t = getattr(m, attr)
                    if torch.is_tensor(t):
                        named.append(('%d/%s.%s' % (b, lname, attr), t))
        loss = sum(torch.norm(t, p='fro') for _, t in named)
        t1 = time.time()
        loss.backward()
        print('%s / %s: reference forward %.1f s, backward %.1f s, loss %.6f' % (case, variant, t1 - t0, time.time() - t1,
                                                                               float(loss)))
        out = {'meta/variant': np.asarray([variant]), 'meta/nodes': np.asarray(nodes, dtype=np.int64),
               'meta/seed0': np.asarray([seed0], dtype=np.int64), 'meta/weights_seed': np.asarray([BENCH_SEED], dtype=np.int64),
               'meta/loss': np.asarray([float(loss)], dtype=np.float64)}
        total = 0
        for name, t in named:
            q = t.detach().reshape(-1)
            total += q.numel()
            idx = recipe.sample_indices(q.numel(), BENCH_SAMPLES, seed=len(name))
            out['pred/%s/norm' % name] = np.asarray([float(q.double().norm())])
            out['pred/%s/sample' % name] = q[idx].numpy().astype(np.float32)
        out['meta/n_predicted'] = np.asarray([total], dtype=np.int64)
        for name, p in ghn.named_parameters():
            gq = p.grad.detach().reshape(-1)
            idx = recipe.sample_indices(gq.numel(), BENCH_SAMPLES, seed=len(name))
            out['grad/%s/norm' % name] = np.asarray([float(gq.double().norm())])
            out['grad/%s/sample' % name] = gq[idx].numpy().astype(np.float32)
        np.savez_compressed(os.path.join(HERE, 'bench_%s_%s.npz' % (case, variant)), **out)
        print('bench_%s_%s.npz: %d predicted params, %d GHN parameter gradients' % (case, variant, total,
                                                                                len(list(ghn.named_parameters()))))
        del ghn, named, loss


# ------------------------------------------------------------------------------------------------
# 5. DeepNets-1M records through the reference's own ``DeepNets1MDDP._init_graph`` (deepnets1m.py:155-279): stored
#    adjacency + node ids (written by ghn3_amd.deepnets1m_io.record_from_graph from sampled architectures of the search
#    space, in the new and the old naming, clean and with the two stored-graph defects injected) -> node features,
#    repaired adjacency, node_info.  The base class (ppuda DeepNets1M) is not needed by the method: the instance is made
#    with __new__ and given the attributes the method reads.
# ------------------------------------------------------------------------------------------------

def deepnets1m_cases():
    """Inputs of the cases (shared by this script and tests/test_deepnets1m_cpu.py): name -> (adj, node triples, net_args)."""
    from ghn3_amd import deepnets1m as D, ops
    from ghn3_amd.graph import Graph as MyGraph
    from ghn3_amd.deepnets1m_io import record_from_graph
    nets = D.SampledNets(num_nets=1000, seed=11, max_nodes=400)
    cases, seen = {}, set()
    want = {('stem1', 'bn'), ('stem0', 'bn'), ('vit', 'bn')}
    for idx in range(200):
        g = nets[idx]
        a = g.net_args
        vit = any(n[0] == 'msa' for n in list(a['genotype'].normal) + list(a['genotype'].reduce))
        kind = ('vit' if vit else 'stem%d' % a['stem_type'], a['norm'])
        if kind not in want or (kind in seen and len([c for c in cases if c.startswith(kind[0])]) >= 4):
            continue
        seen.add(kind)
        model = ops.Network(**a)
        built = MyGraph(model, ve_cutoff=50)
        for old in (False, True):
            adj, triples = record_from_graph(built, old_names=old)
            tag = '%s_%d%s' % (kind[0], idx, '_old' if old else '')
            cases[tag] = (adj, triples, a)
            if old:
                continue
            direct = (adj == 1)
            # defect 1 (stem_type 1): the second cell hangs on stem0's last node instead of stem1's
            if kind[0] == 'stem1':
                # (the stored graphs list the stem's nodes first -- stem1's last node is node 6, deepnets1m.py:170 -- while the
                # generation order of Graph(model) puts the first cell's preprocessing convolutions in front of it: reorder)
                s0 = 4
                s1 = [k for k, t in enumerate(triples) if t[2] == 'stem1.2.weight'][0]
                perm = list(range(6)) + [s1] + [k for k in range(6, len(triples)) if k != s1]
                adj_p, tri_p = adj[perm][:, perm], [triples[k] for k in perm]
                assert tri_p[4][2] == 'stem0.4.weight' and tri_p[6][2] == 'stem1.2.weight' and \
                    not np.tril(adj_p == 1).any()
                cases[tag + '_stemorder'] = (adj_p.copy(), tri_p, a)
                outs1 = np.nonzero(adj_p[6] == 1)[0]
                if len(outs1) == 2:
                    bad = adj_p.copy()
                    bad[bad > 1] = 0
                    bad[6, outs1[-1]] = 0
                    bad[s0, outs1[-1]] = 1
                    cases[tag + '_stemdefect'] = (_with_ve(bad), tri_p, a)
            # defect 2: a layer with two producers
            prims = [t[0] for t in triples]
            cand = [k for k in range(8, len(prims)) if prims[k].startswith(('conv_', 'bn', 'sep_conv', 'dil_conv'))
                    and direct[:, k].sum() == 1]
            if cand:
                k = cand[len(cand) // 2]
                src = [j for j in range(k - 1) if not direct[j, k] and prims[j] != 'input'][-3]
                bad = adj.copy()
                bad[bad > 1] = 0
                bad[src, k] = 1
                cases[tag + '_twoproducers'] = (_with_ve(bad), triples, a)
        if len(cases) >= 24:
            break
    return cases


def _with_ve(A, cutoff=50):
    from ghn3_amd.graph_build import _virtual_edges
    return _virtual_edges(np.array(A, dtype=np.int64), cutoff)


def make_deepnets1m_goldens():
    install_standins()
    sys.path.insert(0, REF)
    import ghn3                                              # noqa: F401
    from ghn3.deepnets1m import DeepNets1MDDP
    out = {}
    for tag, (adj, triples, a) in deepnets1m_cases().items():
        prims, names = {}, {}
        ids = np.zeros((len(triples), 3), dtype=np.int64)
        for k, (ext, cell, name) in enumerate(triples):
            ids[k] = (prims.setdefault(ext, len(prims)), cell, names.setdefault(name, len(names)))
        ds = DeepNets1MDDP.__new__(DeepNets1MDDP)
        ds.debug, ds.dense, ds.virtual_edges = False, True, 50
        ds.primitives_ext = [n for n, _ in sorted(prims.items(), key=lambda kv: kv[1])]
        ds.op_names_net = [n for n, _ in sorted(names.items(), key=lambda kv: kv[1])]
        ds.primitives_dict = {op[:4]: i for i, op in enumerate(ppuda_base.PRIMITIVES_DEEPNETS1M)}
        net_args = {k: v for k, v in a.items() if k not in ('is_imagenet_input', 'num_classes')}
        g = ds._init_graph(adj.copy(), ids, net_args)
        out[tag + '/node_feat'] = g.node_feat.view(-1).numpy().astype(np.int16)
        out[tag + '/A'] = g._Adj.numpy().astype(np.int16)
        out[tag + '/node_info'] = np.asarray([repr([(int(q[0]), str(q[1]), str(q[2]), None if q[3] is None else tuple(
            int(v) for v in q[3]), bool(q[4]), bool(q[5])) for q in cell]) for cell in g.node_info])
        out[tag + '/shapes'] = np.asarray([repr(None if s_ is None else tuple(int(v) for v in s_)) for s_ in g._param_shapes])
        out[tag + '/in_adj_crc'] = np.asarray([int(np.asarray(adj, dtype=np.int64).sum()), int((adj == 1).sum())])
        print(tag, 'nodes', len(triples), 'repaired edges', int((g._Adj.numpy() != np.minimum(adj, 50)).sum()))
    np.savez_compressed(os.path.join(HERE, 'deepnets1m_cases.npz'), **out)
    print('deepnets1m_cases.npz: %d arrays' % len(out))


if __name__ == '__main__':
    # python make_golden.py            -> tiny fixtures (seconds)
    # python make_golden.py graphs     -> graphs.npz (reference Graph(model) on tests/golden/graph_nets.py)
    # python make_golden.py extra      -> ghn3_tiny_extra.npz (big kernels, weight_norm / layernorm off)
    # python make_golden.py networks   -> networks.npz (reference Network / NetworkLight on tests/golden/network_cases.py)
    # python make_golden.py resnet     -> + ResNet-18 / ghn3tm8 and ResNet-50 / ghn3xlm16 (minutes, ~10 GB of RAM)
    # python make_golden.py deepnets1m -> deepnets1m_cases.npz (reference DeepNets1MDDP._init_graph on stored-format records)
    # python make_golden.py bench    -> bench_b1 / bench_b2r: the benchmarked ghn3xlm16 workload, forward + backward (minutes)
    torch.set_num_threads(8 if ('resnet' in sys.argv[1:] or 'bench' in sys.argv[1:]) else 4)
    if 'deepnets1m' in sys.argv[1:]:
        make_deepnets1m_goldens()
    elif 'bench' in sys.argv[1:]:
        make_bench_goldens([a for a in sys.argv[1:] if a in BENCH_CASES] or tuple(BENCH_CASES))
    elif 'graphs' in sys.argv[1:]:
        make_graph_goldens()
    elif 'extra' in sys.argv[1:]:
        make_extra_goldens()
    elif 'networks' in sys.argv[1:]:
        make_network_goldens()
    elif 'resnet' in sys.argv[1:]:
        make_resnet_goldens([a for a in sys.argv[1:] if a in ('18', '50', 'vit')] or ('18', '50', 'vit'))
    else:
        make_graphormer_goldens()
        make_ghn3_goldens()
