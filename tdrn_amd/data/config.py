"""Prior-box configuration dicts with the schema of data/config.py:57-81 of the reference (only
the refinement-detector entries the inference path uses; dataset roots are out of scope)."""


def _refine_cfg(name, min_dim, feature_maps):
    return {
        'feature_maps': list(feature_maps),
        'min_dim': min_dim,
        'steps': [8, 16, 32, 64],
        'min_sizes': [32, 64, 128, 256],
        'max_sizes': [],
        'aspect_ratios': [[2], [2], [2], [2]],
        'variance': [0.1, 0.2],
        'clip': True,
        'flip': True,
        'name': name,
    }


VOC_320 = _refine_cfg('VOC_320', 320, (40, 20, 10, 5))
VOC_512_RefineDet = _refine_cfg('VOC_512_RefineDet', 512, (64, 32, 16, 8))

mb_cfg = {'VOC_320': VOC_320, 'VOC_512_RefineDet': VOC_512_RefineDet}

# multi-scale testing (data/config.py:139-261): the same four-level layout at every input size
VOC_192 = _refine_cfg('VOC_192', 192, (24, 12, 6, 3))
VOC_384 = _refine_cfg('VOC_384', 384, (48, 24, 12, 6))
VOC_448 = _refine_cfg('VOC_448', 448, (56, 28, 14, 7))
VOC_512_s = _refine_cfg('VOC_512_s', 512, (64, 32, 16, 8))
VOC_576 = _refine_cfg('VOC_576', 576, (72, 36, 18, 9))
VOC_704 = _refine_cfg('VOC_704', 704, (88, 44, 22, 11))
VOC_512_RefineDet_06 = _refine_cfg('VOC_512_RefineDet_06', 320, (40, 20, 10, 5))
VOC_512_RefineDet_12 = _refine_cfg('VOC_512_RefineDet_12', 640, (80, 40, 20, 10))
VOC_512_RefineDet_22 = _refine_cfg('VOC_512_RefineDet_22', 1216, (152, 76, 38, 19))
multi_cfg = {'192': VOC_192, '320': VOC_320, '384': VOC_384, '448': VOC_448, '512': VOC_512_s, '576': VOC_576,
             '704': VOC_704}
multi_cfg_512 = {'320': VOC_512_RefineDet_06, '512': VOC_512_RefineDet, '640': VOC_512_RefineDet_12,
                 '1216': VOC_512_RefineDet_22}
