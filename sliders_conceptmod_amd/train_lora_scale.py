"""Image-slider trainer for SD-1.x (reference: trainscripts/imagesliders/train_lora-scale.py:42-520): the SD-XL trainer's
two-sided step (train_lora_scale_xl.py) with the four-prompt SD-1.x conditioning -- `predict_noise` on
concat(unconditional, positive) at slider +s for the `high` image and concat(unconditional, neutral) at -s for the `low`
one (:283-318), image pairs resized to 256 x 256 (:219), `.pt` output.  Same CLI flags as the reference script.

    python -m sliders_conceptmod_amd.train_lora_scale --name eyesize --rank 4 --alpha 1 --config_file data/config.yaml \
           --folder_main datasets/eyesize --folders "bigsize, smallsize" --scales "1, -1" """
from . import train_lora_scale_xl
from .train_lora_scale_xl import build_parser, train  # noqa: F401  (re-exported: the reference script's names)


def main(args, models=None, vae=None):
    """Same flags, launch rules (torch.distributed.run -> one rank per GPU over RCCL) and code path as the SD-XL trainer,
    with the SD-1.x conditioning / 256 x 256 pairs / [1, max - 1) timestep draw selected by xl=False."""
    return train_lora_scale_xl.main(args, xl=False, models=models, vae=vae)


if __name__ == "__main__":
    p = build_parser()
    p.set_defaults(config_file="data/config.yaml")
    main(p.parse_args())
