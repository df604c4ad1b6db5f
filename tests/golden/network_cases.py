"""Target-network cases of tests/golden/networks.npz: genotype, Network keyword arguments, image batch shape.
Data only (hand-written for this repository): every op of the search space and every stem / head variant appears."""

_CONV = dict(
    normal=[('sep_conv_3x3', 0), ('skip_connect', 1), ('conv_1x1', 1), ('max_pool_3x3', 0),
            ('dil_conv_5x5', 2), ('cse', 1), ('conv_7x1_1x7', 3), ('none', 2)],
    normal_concat=[2, 3, 4, 5],
    reduce=[('max_pool_3x3', 0), ('sep_conv_5x5', 1), ('skip_connect', 2), ('avg_pool_3x3', 0),
            ('conv_3x3', 1), ('skip_connect', 2), ('dil_conv_3x3', 0), ('conv_5x5', 3)],
    reduce_concat=[2, 3, 4, 5])

_PLAIN = dict(normal=[('conv_3x3', 0), ('none', 1)], normal_concat=[2],
              reduce=[('conv_3x3', 0), ('none', 1)], reduce_concat=[2])

_VIT = dict(normal=[('msa', 0), ('skip_connect', 1)], normal_concat=[2],
            reduce=[('msa', 0), ('skip_connect', 1)], reduce_concat=[2])

_NONE_STATE = dict(normal=[('none', 0), ('none', 1), ('sep_conv_3x3', 0), ('avg_pool_3x3', 1)], normal_concat=[2, 3],
                   reduce=[('none', 0), ('none', 1), ('max_pool_3x3', 0), ('conv_1x1', 1)], reduce_concat=[2, 3])

CASES = {
    # CIFAR-style input, simple stem, DARTS-like cells with every conv / pool / cse op, auxiliary head
    'cifar_darts': (_CONV, dict(C=8, num_classes=10, n_cells=5, is_imagenet_input=False, norm='bn', auxiliary=True),
                    (4, 3, 32, 32)),
    # ImageNet-style stem (stem0 / stem1, FactorizedReduce preprocessing), pooled stem off, two fc layers
    'imagenet_stem1': (_CONV, dict(C=8, num_classes=12, n_cells=3, is_imagenet_input=True, stem_type=1, norm='bn',
                                   fc_layers=2, fc_dim=16), (2, 3, 64, 64)),
    # simple ImageNet stem with max-pool, no global pooling is not possible at 64 x 64: glob_avg on, stride 2
    'imagenet_stem0_pool': (_CONV, dict(C=8, num_classes=12, n_cells=4, is_imagenet_input=True, stem_pool=True,
                                        imagenet_stride=2, norm='bn', ks=5), (2, 3, 64, 64)),
    # plain chain without preprocessing layers: Stride / Identity preprocessing, widening before a reduction
    'plain_nopreproc': (_PLAIN, dict(C=8, num_classes=10, n_cells=4, is_imagenet_input=False, norm='bn', preproc=False,
                                     C_mult=1), (4, 3, 32, 32)),
    # no normalisation layers at all
    'plain_nonorm': (_PLAIN, dict(C=8, num_classes=10, n_cells=3, is_imagenet_input=False, norm=None, preproc=False,
                                  C_mult=1), (4, 3, 32, 32)),
    # vision transformer: conv_stride patch stem, positional encoding, msa cells
    'vit': (_VIT, dict(C=32, num_classes=10, n_cells=3, is_imagenet_input=False, norm='bn', preproc=False, C_mult=1),
            (4, 3, 32, 32)),
    # a state that receives nothing ('none' twice) and is replaced by zeros in the concatenation
    'none_state': (_NONE_STATE, dict(C=8, num_classes=10, n_cells=3, is_imagenet_input=False, norm='bn'),
                   (4, 3, 32, 32)),
}
