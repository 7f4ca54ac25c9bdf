"""`Cfg` — the reference's flat class-attribute configuration (new_scripy.py:22-67), same names and
defaults, mutable in place (the reference reads it at call time inside DDPM.forward and as default
arguments).  `Config` is the README's spelling of the same class.  New attributes carry
reference-preserving defaults: BOTTLENECK_K (AvgPool / ConvTranspose kernel of the bottleneck, the
reference hard-codes 8), DTYPE (compute dtype of the HIP path) and WORLD_SIZE.
"""
import torch


class Cfg:
    # model
    N_FEAT = 192
    IN_CH = 3
    N_T = 700
    BETAS = (1e-4, 0.02)
    DROP_PROB = 0.1
    # attention-mask thresholds / loss weights (new_scripy.py:30-36)
    HIGH_THRESH = 1.2
    MID_THRESH = 0.8
    HIGH_WEIGHT = 3.0
    MID_WEIGHT = 1.0
    LOW_WEIGHT = 0.5
    FEAT_CONSIST_WEIGHT = 2.0
    # training
    BATCH_SIZE = 4
    ACCUM_STEPS = 4
    LR = 1e-4
    WD = 1e-5
    N_EPOCH = 400
    SAVE_FREQ = 50
    MIN_SAVE_EP = 200
    PATIENCE = 10
    MIN_DELTA = 0.001
    # dataset
    VAL_SPLIT = 0.1
    NUM_WORKERS = 5
    PIN_MEM = True
    # output
    SAVE_DIR = './output/diffusion/'
    SAMPLE_DIR = './output/samples/'
    # sampling
    GUIDE_SCALES = [2.0, 4.0]
    SAMPLES_PER_CLASS = 3
    # image
    IMG_SIZE = 256
    NORM_MEAN = (0.5, 0.5, 0.5)
    NORM_STD = (0.5, 0.5, 0.5)
    # ---- additions of this implementation (defaults keep the reference's behaviour)
    BOTTLENECK_K = 8
    DTYPE = "float32"          # "float32" (1e-4 parity mode), "bfloat16" (throughput mode) or "float16" (the reference's autocast dtype, with loss scaling)
    WORLD_SIZE = 1

    @classmethod
    def torch_dtype(cls):
        return {"float32": torch.float32, "fp32": torch.float32, "bfloat16": torch.bfloat16, "bf16": torch.bfloat16,
                "float16": torch.float16, "fp16": torch.float16, "half": torch.float16}[cls.DTYPE]

    @classmethod
    def loss_constants(cls):
        return [cls.HIGH_THRESH, cls.MID_THRESH, cls.HIGH_WEIGHT, cls.MID_WEIGHT, cls.LOW_WEIGHT, cls.FEAT_CONSIST_WEIGHT]


Config = Cfg
