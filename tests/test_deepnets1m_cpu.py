"""
DeepNets-1M on-disk format, graph repairs and per-item arguments (/root/reference/ghn3/deepnets1m.py:84-319)
against goldens produced by the reference's own ``DeepNets1MDDP._init_graph`` (tests/golden/make_golden.py deepnets1m ->
tests/golden/deepnets1m_cases.npz).  CPU only.
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
sys.path.insert(0, os.path.dirname(HERE))

from ghn3_amd import deepnets1m_io as io                    # noqa: E402
from ghn3_amd import ops                                     # noqa: E402


@pytest.fixture(scope='module')
def cases():
    import make_golden
    return make_golden.deepnets1m_cases()


@pytest.fixture(scope='module')
def golden():
    return np.load(os.path.join(HERE, 'golden', 'deepnets1m_cases.npz'))


def _ids(triples):
    prims, names = {}, {}
    ids = np.zeros((len(triples), 3), dtype=np.int64)
    for k, (ext, cell, name) in enumerate(triples):
        ids[k] = (prims.setdefault(ext, len(prims)), cell, names.setdefault(name, len(names)))
    return ids, [n for n, _ in sorted(prims.items(), key=lambda kv: kv[1])], \
        [n for n, _ in sorted(names.items(), key=lambda kv: kv[1])]


def _info_repr(node_info):
    return [repr([(int(q[0]), str(q[1]), str(q[2]), None if q[3] is None else tuple(int(v) for v in q[3]), bool(q[4]),
                   bool(q[5])) for q in cell]) for cell in node_info]


def test_init_graph_matches_the_reference_method(cases, golden):
    """Every case (clean records in the new and the old naming, the stem defect of stem_type = 1, a layer with two producers,
    ViT cells): node features, repaired adjacency (virtual edges recomputed), node_info and shapes are the reference's."""
    tags = sorted({k.split('/')[0] for k in golden.files})
    assert set(tags) == set(cases) and len(tags) >= 15
    assert any(t.endswith('_stemdefect') for t in tags) and any(t.endswith('_twoproducers') for t in tags)
    for tag in tags:
        adj, triples, a = cases[tag]
        assert [int(adj.sum()), int((adj == 1).sum())] == golden[tag + '/in_adj_crc'].tolist(), tag   # same input as the golden run
        ids, prims, names = _ids(triples)
        net_args = {k: v for k, v in a.items() if k not in ('is_imagenet_input', 'num_classes')}
        g = io.init_graph(adj.copy(), ids, net_args, prims, names, virtual_edges=50)
        assert np.array_equal(g.node_feat.view(-1).numpy(), golden[tag + '/node_feat']), tag
        assert np.array_equal(g._Adj.numpy(), golden[tag + '/A']), tag
        assert _info_repr(g.node_info) == golden[tag + '/node_info'].tolist(), tag
        assert [repr(None if s is None else tuple(int(v) for v in s)) for s in g._param_shapes] == \
            golden[tag + '/shapes'].tolist(), tag


def test_repairs_restore_the_network_wiring(cases, golden):
    """The stem repair turns the defective record back into the graph of the network itself; the old naming normalises to
    the names GHN3.forward matches against the network's parameter table."""
    for tag in [t for t in cases if t.endswith('_stemdefect')]:
        clean = tag.replace('_stemdefect', '_stemorder')
        assert np.array_equal(golden[tag + '/A'], golden[clean + '/A'])
        assert not np.array_equal(np.minimum(cases[tag][0], 50), golden[tag + '/A'])
    for tag in [t for t in cases if t.endswith('_old')]:
        assert golden[tag + '/node_info'].tolist() == golden[tag[:-4] + '/node_info'].tolist()
        adj, triples, a = cases[tag[:-4]]
        net = ops.NetworkLight(**a)
        table = [set(cell) for cell in net._layered_modules]
        for c, cell in enumerate(eval(r) for r in golden[tag + '/node_info'].tolist()):
            for (_, name, prim, _, _, _) in cell:
                if prim not in ('max_pool', 'avg_pool'):
                    assert name in table[c], (tag, c, name)


def test_file_round_trip_and_dataset_items(tmp_path, cases):
    """Writer -> (hdf5 | npz) + meta json -> NetStore / DeepNets1MDDP: the training item carries the repaired graph, fresh
    width draws inside the reference's ranges and a NetworkLight whose parameter table the GHN's host compiler accepts;
    evaluation items keep the stored widths; the sampler skips over-budget meta-batches like deepnets1m.py:298-317."""
    from ghn3_amd import DeepNets1MDDP, NetBatchSamplerDDP, GraphBatch
    from ghn3_amd.program import Program
    w = io.Writer()
    keep = [t for t in sorted(cases) if t.count('_') == 1][:5]
    for split in ('train', 'val'):
        for tag in keep:
            adj, triples, a = cases[tag]
            w.add(split, a, adj, triples, num_params={'cifar10': 1.1e5, 'imagenet': 2.2e5})
    path = w.save(str(tmp_path), 'train')
    if w.splits.get('val'):
        w2 = io.Writer()
        w2.prims, w2.names, w2.splits = w.prims, w.names, {'val': w.splits['val']}
        w2.save(str(tmp_path), 'val')
    assert os.path.exists(path)
    ds = DeepNets1MDDP(split='train', nets_dir=str(tmp_path), virtual_edges=50)
    assert ds.store is not None and len(ds) == len(keep) and ds.nodes.tolist() == [len(cases[t][1]) for t in keep]
    torch.manual_seed(3)
    np.random.seed(3)
    items = [ds[i] for i in range(len(ds))]
    torch.manual_seed(3)
    np.random.seed(3)
    again = [ds[i] for i in range(len(ds))]
    for tag, g, g2 in zip(keep, items, again):
        adj, triples, a = cases[tag]
        assert g.net_args['C'] == g2.net_args['C'] and g.net_args['fc_dim'] == g2.net_args['fc_dim']   # seeded draws
        assert g.net_args['C'] in io.NUM_CH.tolist() and g.net_args['fc_dim'] in io.FC_DIM.tolist()
        assert g.net_args['imagenet_stride'] == 4 and g.n_nodes == len(triples)
        assert isinstance(g.net, ops.NetworkLight) and g.net_idx == keep.index(tag)
    gb = GraphBatch(items[:2], dense=True)
    gb._cat()
    cfg = dict(hid=32, heads=4, layers=2, num_classes=10, max_shape=(32, 32, 16, 16))
    # (every parameter of the attached networks is matched by a node of the repaired graph: GHN3.forward's bookkeeping,
    # nn.py:594-692, predicts exactly the networks' parameter count; reduce_graph=True consumes the tables it matched)
    n_table = sum(int(np.prod(e['sz'])) for n in gb.nets for cell in n._layered_modules for e in cell.values())
    prog = Program(cfg, gb.node_info, gb.host_n_nodes(), gb._node_type_host, gb.max_edge, gb.nets, training=True,
                   reduce_graph=True)
    assert sum(p['numel'] for p in prog.predicted) == n_table
    ev = DeepNets1MDDP(split='val', nets_dir=str(tmp_path), virtual_edges=50)
    g = ev[0]
    assert not hasattr(g, 'net') and g.net_args['C'] == cases[keep[0]][2]['C']
    # node budget: with the stored counts the SAMPLER skips (the reference's rule), the collate function truncates nothing
    loader, sampler = DeepNets1MDDP.loader(meta_batch_size=2, split='train', nets_dir=str(tmp_path), num_workers=0)
    assert isinstance(sampler, NetBatchSamplerDDP)
    sampler.max_nodes_batch = int(np.sort(ds.nodes)[:2].sum())          # only the two smallest graphs fit together
    it = iter(sampler)
    for _ in range(6):
        batch = next(it)
        assert int(ds.nodes[batch].sum()) <= sampler.max_nodes_batch
    gb = next(iter(loader))
    assert gb.dropped_graphs == 0 and len(gb.nets) == len(gb)


def test_item_net_args_ranges():
    """deepnets1m.py:99-133: large / dense / deep networks get the narrowest width; 'wide' evaluation widens C."""
    g = ops.Genotype(normal=[('conv_5x5', 0), ('skip_connect', 1)], normal_concat=[2], reduce=[('max_pool_3x3', 0),
                     ('skip_connect', 1)], reduce_concat=[2])
    base = dict(n_cells=6, num_params={'cifar10': 3e5, 'imagenet': 6e5}, glob_avg=True, stem_type=0, stem_pool=True,
                norm='bn', ks=3, preproc=True, C_mult=2, fc_layers=1, C=48, fc_dim=128)
    torch.manual_seed(0)
    for _ in range(20):
        a = io.item_net_args(base, g, True, False, True, 'train')
        assert a['C'] == 32 and a['fc_dim'] in (64, 128, 192, 256)               # conv_5x5: "conv dense" -> min width
    g2 = g._replace(normal=[('sep_conv_3x3', 0), ('skip_connect', 1)])
    seen = set()
    for _ in range(200):
        a = io.item_net_args(dict(base, num_params={'cifar10': 1e5, 'imagenet': 1e5}), g2, True, False, True, 'train')
        seen.add(a['C'])
    assert seen == set(io.NUM_CH.tolist())
    a = io.item_net_args(base, g2, False, True, True, 'wide')
    assert a['C'] == 96 and a['fc_dim'] == 128 and a['imagenet_stride'] == 4
    a = io.item_net_args(base, g2, False, False, True, 'wide')
    assert a['C'] == 192


def test_unreadable_dataset_files_raise_instead_of_falling_back(tmp_path, monkeypatch):
    """A `nets_dir` that holds SOME DeepNets-1M files which cannot be read (the .hdf5 without h5py, the arrays without the
    _meta.json) must not silently become the sampled stand-in stream: the reference raises when it cannot open its files
    (deepnets1m.py:38-46,90-91).  An empty directory still selects the sampled stream, with a warning."""
    import warnings
    from ghn3_amd import deepnets1m_io as io
    from ghn3_amd.deepnets1m import DeepNets1MDDP
    base = os.path.join(str(tmp_path), io.split_file('val'))
    open(base + '.hdf5', 'wb').write(b'not really hdf5')
    with pytest.raises(FileNotFoundError, match='_meta.json'):
        DeepNets1MDDP(split='val', nets_dir=str(tmp_path))
    open(base + '_meta.json', 'w').write('{}')
    monkeypatch.setattr(io, 'h5py', None)
    with pytest.raises(FileNotFoundError, match='h5py'):
        DeepNets1MDDP(split='val', nets_dir=str(tmp_path))
    empty = tmp_path / 'empty'
    empty.mkdir()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        ds = DeepNets1MDDP(split='val', nets_dir=str(empty), num_nets=3)
    assert ds.store is None and len(ds) == 3 and any('sampled architecture stream' in str(x.message) for x in w)
    with pytest.raises(FileNotFoundError, match='arch='):
        DeepNets1MDDP(split='val', nets_dir=str(empty), arch=0)
