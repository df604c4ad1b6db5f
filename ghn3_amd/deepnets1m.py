"""
Training architectures for the GHN (role of /root/reference/ghn3/deepnets1m.py:29-319).

The reference trains on DeepNets-1M: one million architectures stored as (genotype, network arguments) in a json meta
file with their precomputed computational graphs in an hdf5 file, served by ``DeepNets1MDDP.loader`` as ``GraphBatch``
objects that carry a ``NetworkLight`` per graph.  Neither the files nor ``h5py`` exist in this image, so this module
draws architectures from the same search space instead (same op vocabulary, cell structure and argument ranges; the
exact sampling frequencies of the DeepNets-1M generator are NOT reproduced) and builds their graphs with
``ghn3_amd.Graph(model)`` -- which is what generated the graphs of DeepNets-1M in the first place:

    queue = SampledNets.loader(meta_batch_size=8, num_nets=10 ** 6, large_images=False, seed=0)
    for graphs in queue:                      # GraphBatch: .nets = [NetworkLight], dense tensors for GHN3.forward
        trainer.update(images, targets, graphs=graphs)

Every architecture is a pure function of (seed, index): all ranks of a data-parallel job can address the same virtual
dataset and take disjoint slices of each meta-batch (``rank`` / ``world_size``), like the reference's
DistributedSampler (deepnets1m.py:289).
"""

import numpy as np
import torch

from . import ops
from .graph import Graph, GraphBatch

# op vocabulary of the DeepNets-1M generator (ppuda genotypes): name -> available kernel sizes
_SIZED = {'sep_conv': (3, 5, 7), 'dil_conv': (3, 5), 'conv': (1, 3, 5, 7), 'max_pool': (3,), 'avg_pool': (3,)}
_PLAIN = ('skip_connect', 'cse', 'none')


def _op_name(rng, vit):
    kinds = list(_SIZED) + list(_PLAIN) + (['msa'] if vit else []) + ['conv2']
    kind = kinds[rng.randint(len(kinds))]
    if kind == 'conv2':
        return 'conv_7x1_1x7'
    if kind in _SIZED:
        k = _SIZED[kind][rng.randint(len(_SIZED[kind]))]
        return '%s_%dx%d' % (kind, k, k)
    return kind


def sample_genotype(rng, vit=False, max_steps=4):
    """A random cell pair: `steps` intermediate states, two (op, input) pairs each; inputs are earlier states."""
    def cell(steps):
        pairs = []
        for s in range(steps):
            for _ in range(2):
                pairs.append((_op_name(rng, vit), int(rng.randint(s + 2))))
            if all(p[0] == 'none' for p in pairs[-2:]):      # keep every state alive
                pairs[-1] = ('skip_connect', pairs[-1][1])
        used = {i for name, i in pairs if name != 'none'}     # a state read by 'none' only still has to reach the output
        concat = [k for k in range(2, steps + 2) if k not in used] or [steps + 1]
        return pairs, concat
    steps = int(rng.randint(1, max_steps + 1))
    normal, n_cat = cell(steps)
    reduce, r_cat = cell(steps)
    if len(r_cat) != len(n_cat):                             # (the channel bookkeeping assumes equal multipliers)
        reduce, r_cat = normal, n_cat
    return ops.Genotype(normal=normal, normal_concat=n_cat, reduce=reduce, reduce_concat=r_cat)


def sample_net_args(rng, large_images=False):
    """Network keyword arguments in the ranges the DeepNets-1M loader uses (deepnets1m.py:96-143)."""
    vit = rng.rand() < 0.1
    genotype = sample_genotype(rng, vit=vit, max_steps=1 if vit else 4)
    steps = len(genotype.normal_concat)
    preproc = True if (steps > 1) else bool(rng.rand() < 0.5)
    n_cells = int(rng.randint(3, 10)) if not vit else int(rng.randint(3, 7))
    args = dict(genotype=genotype, n_cells=n_cells, C=int(rng.choice([32, 48, 64] if not vit else [32, 64, 128])),
                ks=int(rng.choice([3, 5, 7])), norm='bn', preproc=preproc, C_mult=2 if preproc else 1,
                stem_type=int(rng.randint(2)) if not vit else 0, stem_pool=bool(rng.rand() < 0.5),
                glob_avg=True, fc_layers=int(rng.randint(1, 3)), fc_dim=int(rng.choice([64, 128, 256])),
                imagenet_stride=4)
    if vit:
        args.update(preproc=False, C_mult=1, stem_pool=False)
    args['is_imagenet_input'] = bool(large_images)
    args['num_classes'] = 1000 if large_images else 10
    return args


class SampledNets:
    """A virtual dataset of architectures: ``self[i]`` is a Graph with ``.net`` (a NetworkLight) and ``.net_args``."""

    def __init__(self, num_nets=10 ** 6, large_images=False, seed=0, virtual_edges=50, max_nodes=1000, light=True,
                 verbose=False):
        self.num_nets, self.large_images, self.seed = int(num_nets), large_images, int(seed)
        self.virtual_edges, self.max_nodes, self.light, self.verbose = virtual_edges, max_nodes, light, verbose

    def __len__(self):
        return self.num_nets

    def __getitem__(self, idx):
        idx = int(idx) % self.num_nets
        for attempt in range(64):
            rng = np.random.RandomState((self.seed * 1000003 + idx * 64 + attempt) % (2 ** 31 - 1))
            args = sample_net_args(rng, self.large_images)
            try:
                model = ops.Network(**args)
                graph = Graph(model, ve_cutoff=self.virtual_edges, verbose=False)
            except Exception as e:                            # an invalid draw (e.g. a state of shape zero): redraw
                if self.verbose:
                    print('SampledNets: redraw %d/%d (%s)' % (idx, attempt, repr(e)[:80]))
                continue
            if graph.n_nodes > self.max_nodes:
                continue
            # every parameter must be reachable from the output (a cell that ignores one of its inputs leaves the
            # preprocessing layer of that input outside the graph: the GHN would predict nothing for it)
            # (matched cell by cell, as GHN3.forward does: nn.py:612-650)
            if len(graph.node_info) != len(model._layered_modules):
                continue
            complete = True
            for cell_nodes, cell_table in zip(graph.node_info, model._layered_modules):
                in_graph = {n[1] for n in cell_nodes}
                complete &= all(k in in_graph or k.replace('.bias', '.weight') in in_graph for k in cell_table)
            if not complete:
                continue
            out = Graph(node_feat=graph.node_feat, node_info=graph.node_info, A=graph._Adj, net_args=args, net_idx=idx)
            out.net = ops.NetworkLight(**args) if self.light else model
            return out
        raise RuntimeError('no valid architecture for index %d' % idx)

    @staticmethod
    def loader(meta_batch_size=1, dense=True, rank=0, world_size=1, start_step=0, **kwargs):
        """Endless iterator of GraphBatch objects (``DeepNets1MDDP.loader`` of deepnets1m.py:271-319): step s of rank r
        holds the architectures s * meta_batch_size + [r * m, (r + 1) * m) with m = meta_batch_size / world_size."""
        assert dense, 'GHN-3 uses the dense layout'
        assert meta_batch_size % world_size == 0, (meta_batch_size, world_size)
        nets = SampledNets(**kwargs)
        per_rank = meta_batch_size // world_size

        def generate():
            step = int(start_step)
            while True:
                base = step * meta_batch_size + rank * per_rank
                yield GraphBatch([nets[base + k] for k in range(per_rank)], dense=True)
                step += 1
        return generate()


# ---------------------------------------------------------------------------------------------------------------------
# The reference's names for the same roles (ghn3/__init__.py:13, deepnets1m.py:29-79,281-319)
# ---------------------------------------------------------------------------------------------------------------------
MAX_NODES_BATCH = 2200          # ppuda.deepnets1m.loader.MAX_NODES_BATCH (published value; the package is not in this image)


class DeepNets1MDDP(SampledNets, torch.utils.data.Dataset):
    """``DeepNets1MDDP`` of deepnets1m.py:29-153.  Two sources behind the same ``loader`` contract (a training loader comes
    with its infinite, rank-aware batch sampler, an evaluation loader alone; items are ``Graph`` objects with ``.net`` /
    ``.net_args`` / ``.net_idx``, batches are ``GraphBatch`` objects):

      * the DeepNets-1M files, when ``nets_dir`` holds them (``deepnets1m_<split>.hdf5`` -- or the same arrays as ``.npz``
        where h5py is missing -- and ``..._meta.json``; ghn3_amd.deepnets1m_io): ``__getitem__`` follows deepnets1m.py:84-153
        -- per-visit width / stride draws in training, the stored graph repaired by ``init_graph``, a ``NetworkLight``
        attached in training;
      * otherwise a sampled architecture stream of the same search space (``SampledNets``: neither the dataset files nor
        h5py exist in this image); ``split`` then only selects the stream's seed."""

    def __init__(self, split='train', nets_dir=None, virtual_edges=50, num_nets=None, large_images=False, dense=True,
                 wider_nets=True, debug=False, verbose=False, seed=None, max_nodes=1000, light=True, arch=None, **unused):
        assert dense, 'GHN-3 uses the dense layout'
        self.split, self.is_train, self.dense, self.wider_nets, self.debug = split, split == 'train', dense, wider_nets, debug
        self.store = None
        if nets_dir is not None:
            from .deepnets1m_io import NetStore
            store = NetStore(nets_dir, split)
            why = store.partial()
            if not store.exists() and not why and nets_dir != './data':   # deepnets1m.py:38-46: fall back to a local ./data folder
                store = NetStore('./data', split)
                why = store.partial()
            if why:
                raise FileNotFoundError('DeepNets-1M split %r cannot be read: %s' % (split, why))
            if store.exists():
                self.store = store
            else:
                import warnings
                warnings.warn('no DeepNets-1M files for split %r under %r: using the sampled architecture stream of the same '
                              'search space (SampledNets), NOT the dataset' % (split, nets_dir))
            if self.store is None and arch is not None:
                raise FileNotFoundError('arch=%r selects a stored DeepNets-1M network, but no dataset files were found under %r'
                                        % (arch, nets_dir))
        if self.store is not None:
            nets, self.primitives_ext, self.op_names_net = self.store.load_meta()
            self.nets = nets[:len(nets) if num_nets is None else num_nets]
            self.h5_idx = [arch] if arch is not None else None
            self.nodes = np.asarray([net['num_nodes'] for net in self.nets], dtype=np.int64)   # (the sampler's node budget)
            self.num_nets, self.large_images, self.virtual_edges = len(self.nets), large_images, virtual_edges
            self.light, self.verbose, self.max_nodes, self.seed = light, verbose, max_nodes, 0
            return
        if seed is None:                                        # disjoint streams per split
            seed = {'train': 0, 'val': 1, 'test': 2}.get(split, 3)
        if num_nets is None:
            num_nets = 10 ** 6 if self.is_train else 500
        SampledNets.__init__(self, num_nets=num_nets, large_images=large_images, seed=seed, virtual_edges=virtual_edges,
                             max_nodes=max_nodes, light=light, verbose=verbose)

    def __len__(self):
        return self.num_nets if self.store is None or self.h5_idx is None else len(self.h5_idx)

    def __getitem__(self, idx):
        if self.store is None:
            return SampledNets.__getitem__(self, idx)
        from . import deepnets1m_io as io
        args = self.nets[idx if self.h5_idx is None else self.h5_idx[idx]]
        idx = self.h5_idx[idx] if self.h5_idx is not None else idx
        cell = ops.from_dict(args['genotype'])
        net_args = io.item_net_args(args, cell, self.is_train, self.large_images, self.wider_nets, self.split)
        adj, nodes = self.store.get(idx)
        graph = io.init_graph(adj, nodes, net_args, self.primitives_ext, self.op_names_net,
                              virtual_edges=self.virtual_edges, dense=self.dense, debug=self.debug)
        graph.net_idx = idx
        if self.is_train and not self.debug:
            graph.net = ops.NetworkLight(is_imagenet_input=self.large_images, num_classes=1000 if self.large_images else 10,
                                         **net_args)
        return graph

    @staticmethod
    def loader(meta_batch_size=1, dense=True, num_workers=None, **kwargs):
        from functools import partial
        nets = DeepNets1MDDP(dense=dense, **kwargs)
        sampler = NetBatchSamplerDDP(nets, meta_batch_size) if nets.is_train else None
        if num_workers is None:                                 # (deepnets1m.py:74)
            num_workers = (0 if meta_batch_size <= 1 else min(8, max(4, meta_batch_size // 2))) if nets.is_train else 0
        # Node budget of a meta-batch (deepnets1m.py:288-300).  With the dataset files the node counts are in the meta file
        # and the SAMPLER skips an over-budget meta-batch, exactly as the reference does; the sampled stream has no
        # precomputed counts, so there the collate function enforces the budget once the graphs exist (collate_capped).
        cap = sampler.max_nodes_batch if (sampler is not None and nets.store is None) else None
        loader = torch.utils.data.DataLoader(nets, batch_sampler=sampler, batch_size=1, pin_memory=False,
                                             collate_fn=partial(collate_capped, dense=dense, max_nodes_batch=cap),
                                             num_workers=num_workers)
        return (loader, sampler) if nets.is_train else loader   # (the sampler is returned for distributed training)


TRUNCATED_BATCHES = [0, 0]          # [meta-batches truncated by collate_capped, graphs dropped] in this process (loader worker)


def collate_capped(graphs, dense=True, max_nodes_batch=None):
    """GraphBatch of a meta-batch under the node budget of deepnets1m.py:288-300.  The reference SKIPS a meta-batch whose
    precomputed node counts exceed the budget (and this loader does the same when it reads the dataset files: the sampler
    has the counts).  The sampled stream has no precomputed counts -- graphs are built by the loader workers -- so there
    the budget is enforced here, after the graphs exist: graphs are dropped from the end of the meta-batch until it fits
    (at least one is kept).  DEVIATION from the reference: the effective meta-batch of such a step is smaller (the trainer
    divides by the number of networks it got); truncations are counted in TRUNCATED_BATCHES and on the batch
    (``dropped_graphs``) so that a run can report them."""
    graphs = list(graphs)
    dropped = 0
    if max_nodes_batch is not None:
        while len(graphs) > 1 and sum(int(g.n_nodes) for g in graphs) > max_nodes_batch:
            graphs.pop()
            dropped += 1
    if dropped:
        TRUNCATED_BATCHES[0] += 1
        TRUNCATED_BATCHES[1] += dropped
    gb = GraphBatch(graphs, dense=dense)
    gb.dropped_graphs = dropped
    return gb


class NetBatchSamplerDDP(torch.utils.data.BatchSampler):
    """deepnets1m.py:281-319: endless sampler of meta-batches.  Every epoch is a permutation of the dataset drawn from
    (seed, epoch) -- the same on all ranks -- of which rank r takes the indices r, r + W, r + 2 W ... (DistributedSampler's
    rule, padded by wrapping around to a multiple of the world size), so the ranks train on disjoint architectures.  The
    node budget ``max_nodes_batch`` is applied by the loader's collate function (``collate_capped``): node counts only
    exist once a worker has built the graphs."""

    def __init__(self, deepnets, meta_batch_size=1, seed=0):
        from .ddp_utils import is_ddp, get_ddp_rank
        self.dataset, self.batch_size, self.drop_last, self.seed = deepnets, int(meta_batch_size), False, int(seed)
        self.rank, self.world = (get_ddp_rank(), torch.distributed.get_world_size()) if is_ddp() else (0, 1)
        self.max_nodes_batch = int(MAX_NODES_BATCH / 8 * max(8, meta_batch_size)) \
            if deepnets.is_train and meta_batch_size > 1 else None
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def epoch_indices(self, epoch):
        n = len(self.dataset)
        if self.dataset.is_train:
            perm = np.random.RandomState((self.seed * 7919 + epoch) % (2 ** 31 - 1)).permutation(n)
        else:
            perm = np.arange(n)
        total = (n + self.world - 1) // self.world * self.world
        perm = np.concatenate([perm, perm[:total - n]])
        return perm[self.rank:total:self.world]

    def __len__(self):
        return (len(self.dataset) // self.world + self.batch_size - 1) // self.batch_size

    def check_batch(self, batch):
        """deepnets1m.py:298-301: with precomputed node counts (dataset files) an over-budget meta-batch is skipped."""
        nodes = getattr(self.dataset, 'nodes', None) if getattr(self.dataset, 'store', None) is not None else None
        return self.max_nodes_batch is None or nodes is None or int(nodes[batch].sum()) <= self.max_nodes_batch

    def __iter__(self):
        epoch = self.epoch
        while True:                                             # infinite sampler
            batch = []
            for idx in self.epoch_indices(epoch):
                batch.append(int(idx))
                if len(batch) == self.batch_size:
                    if self.check_batch(batch):
                        yield batch
                    batch = []
            if len(batch) > 0 and not self.drop_last:
                if self.check_batch(batch):
                    yield batch
            epoch += 1
            if not self.dataset.is_train:
                return
