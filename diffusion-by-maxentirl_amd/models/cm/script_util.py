"""`models.cm.script_util` — model/diffusion factory with the reference keyword surface
(reference: models/cm/script_util.py:22-157).  YAML `diffusion:` blocks written for the reference
instantiate unchanged: create_model_and_diffusion(**cfg.diffusion).
"""
import argparse

from .karras_diffusion import KarrasDenoiser
from .unet import UNetModel

NUM_CLASSES = 1000


def model_and_diffusion_defaults():
    return dict(sigma_min=0.002, sigma_max=80.0, image_size=64, num_channels=128, num_res_blocks=2, num_heads=4,
                num_heads_upsample=-1, num_head_channels=-1, attention_resolutions="32,16,8", channel_mult="",
                dropout=0.0, class_cond=False, use_checkpoint=False, use_scale_shift_norm=True, resblock_updown=False,
                use_fp16=False, use_new_attention_order=False, learn_sigma=False, weight_schedule="karras")


def create_model_and_diffusion(image_size, class_cond, learn_sigma, num_channels, num_res_blocks, channel_mult, num_heads,
                               num_head_channels, num_heads_upsample, attention_resolutions, dropout, use_checkpoint,
                               use_scale_shift_norm, resblock_updown, use_fp16, use_new_attention_order, weight_schedule,
                               sigma_min=0.002, sigma_max=80.0, distillation=False):
    model = create_model(image_size, num_channels, num_res_blocks, channel_mult=channel_mult, learn_sigma=learn_sigma,
                         class_cond=class_cond, use_checkpoint=use_checkpoint, attention_resolutions=attention_resolutions,
                         num_heads=num_heads, num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
                         use_scale_shift_norm=use_scale_shift_norm, dropout=dropout, resblock_updown=resblock_updown,
                         use_fp16=use_fp16, use_new_attention_order=use_new_attention_order)
    diffusion = KarrasDenoiser(sigma_data=0.5, sigma_max=sigma_max, sigma_min=sigma_min, distillation=distillation,
                               weight_schedule=weight_schedule)
    return model, diffusion


def create_model(image_size, num_channels, num_res_blocks, channel_mult="", learn_sigma=False, class_cond=False,
                 use_checkpoint=False, attention_resolutions="16", num_heads=1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, dropout=0, resblock_updown=False, use_fp16=False, use_new_attention_order=False):
    if channel_mult == "":
        table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}
        if image_size not in table:
            raise ValueError(f"unsupported image size: {image_size}")
        channel_mult = table[image_size]
    else:
        channel_mult = tuple(int(m) for m in str(channel_mult).split(","))
    attention_ds = tuple(image_size // int(res) for res in str(attention_resolutions).split(","))   # str(): a CLI override "16" arrives as int
    return UNetModel(image_size=image_size, in_channels=3, model_channels=num_channels,
                     out_channels=(3 if not learn_sigma else 6), num_res_blocks=num_res_blocks,
                     attention_resolutions=attention_ds, dropout=dropout, channel_mult=channel_mult,
                     num_classes=(NUM_CLASSES if class_cond else None), use_checkpoint=use_checkpoint, use_fp16=use_fp16,
                     num_heads=num_heads, num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
                     use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown,
                     use_new_attention_order=use_new_attention_order)


def add_dict_to_argparser(parser, default_dict):
    for k, v in default_dict.items():
        v_type = str if v is None else (str2bool if isinstance(v, bool) else type(v))
        parser.add_argument(f"--{k}", default=v, type=v_type)


def args_to_dict(args, keys):
    return {k: getattr(args, k) for k in keys}


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("boolean value expected")
