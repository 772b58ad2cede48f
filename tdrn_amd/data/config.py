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
