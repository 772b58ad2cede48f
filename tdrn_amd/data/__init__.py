from .config import mb_cfg, VOC_320, VOC_512_RefineDet   # noqa: F401
