"""Shared helpers for the GPU parity tests: build the HIP model and the CPU oracle with identical weights."""

import numpy as np
import torch

import recipe
from oracle import ghn3_ref as R
from oracle import graphormer_ref as G


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / den) if den > 0 else float(np.linalg.norm(a - b))


def make_models(cfg, seed, index_mode='reference', compute='f32', device='cuda'):
    """(hip model on device, oracle model on CPU) sharing one seeded state dict."""
    from ghn3_amd import GHN3
    oracle = R.GHN3Ref(**cfg, index_mode=index_mode)
    shapes = {k: tuple(v.shape) for k, v in oracle.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in recipe.seeded_state_dict(shapes, seed=seed).items()}
    oracle.load_state_dict(sd)
    hip = GHN3(**cfg, index_mode=index_mode, compute=compute)
    hip.load_state_dict(sd)
    hip = hip.to(device)
    return hip, oracle


def tiny_case(case):
    """(nets for HIP, GraphBatch for HIP, nets for oracle, GraphBatchRef) of a committed tiny case."""
    from ghn3_amd import Graph, GraphBatch
    specs = recipe.edge_specs(case) if case in recipe.EDGE_CASES else \
        recipe.EXTRA_CASES[case][0] if case in recipe.EXTRA_CASES else [recipe.TINY_NETS[i] for i in recipe.TINY_CASES[case]]
    nets_h = [recipe.build_torch_net(s) for s in specs]
    nets_o = [recipe.build_torch_net(s) for s in specs]
    gh, go = [], []
    for s in specs:
        nf, info, A = recipe.graph_arrays(s)
        gh.append(Graph(node_feat=nf, node_info=info, A=A))
        go.append(R.GraphRef(torch.from_numpy(nf), info, torch.from_numpy(A)))
    return nets_h, GraphBatch(gh, dense=True), nets_o, R.GraphBatchRef(go)


def synthetic_case(n_nodes_list, seed0):
    """Same synthetic graphs for the HIP model and for the oracle (LightNet targets on both sides)."""
    from ghn3_amd.synthetic import synthetic_batch
    gb_h, nets_h = synthetic_batch(n_nodes_list, seed0)
    gb_o, nets_o = synthetic_batch(n_nodes_list, seed0)
    go = [R.GraphRef(g.node_feat, g.node_info, g._Adj) for g in gb_o.graphs]
    return nets_h, gb_h, nets_o, R.GraphBatchRef(go)


def oracle_intermediates(oracle, nets, batch):
    """x0 after the layer-0 prologue, edge bias, x after every layer, final embeddings."""
    p = dict(oracle.named_parameters())
    pg, pm = R.map_net_params(batch, nets, oracle.max_shape)
    with torch.no_grad():
        xs = oracle.node_embeddings(batch, pm)
        x = batch.to_dense(xs)
        mask = batch.mask & batch.mask.permute(0, 2, 1)
        x0, e2 = G.layer0_prologue(x.clone(), batch.edges, mask, p, 'gnn.0.')
        bias = G.edge_bias(e2, p, 'gnn.0.')
        out = {'x0': x0, 'bias': bias.permute(0, 3, 1, 2).contiguous()}
        xc, b = x, batch.edges
        for l in range(oracle.layers):
            xc, b = G.transformer_layer(xc, b, mask, p, 'gnn.%d.' % l, oracle.heads, layer0=(l == 0))
            out['x%d' % (l + 1)] = xc
        out['xe'] = oracle.graphormer(xs, batch)
    return out


def ws_tensor(plan, name, shape):
    prog = plan.program
    off = prog._ws_names[name]
    n = int(np.prod(shape))
    return plan.ws[off:off + 4 * n].view(torch.float32).view(*shape)


def predicted_dict_hip(plan, flat):
    out = {}
    for k, p in enumerate(plan.program.predicted):
        out[k] = flat[p['offset']:p['offset'] + p['numel']].view(p['tile_shape'])
    return out
