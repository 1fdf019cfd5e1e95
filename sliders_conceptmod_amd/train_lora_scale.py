"""Image-slider trainer for SD-1.x (reference: trainscripts/imagesliders/train_lora-scale.py:42-520): the SD-XL trainer's
two-sided step (train_lora_scale_xl.py) with the four-prompt SD-1.x conditioning -- `predict_noise` on
concat(unconditional, positive) at slider +s for the `high` image and concat(unconditional, neutral) at -s for the `low`
one (:283-318), image pairs resized to 256 x 256 (:219), `.pt` output.  Same CLI flags as the reference script.

    python -m sliders_conceptmod_amd.train_lora_scale --name eyesize --rank 4 --alpha 1 --config_file data/config.yaml \
           --folder_main datasets/eyesize --folders "bigsize, smallsize" --scales "1, -1" """
import torch

from . import config_util, prompt_util
from .train_lora_scale_xl import build_parser, train


def main(args):
    config = config_util.load_config_from_yaml(args.config_file)
    if args.name is not None:
        config.save.name = args.name
    attributes = [a.strip() for a in args.attributes.split(",")] if args.attributes is not None else []
    config.network.alpha = args.alpha
    config.network.rank = args.rank
    config.save.name += f"_alpha{args.alpha}_rank{config.network.rank}_{config.network.training_method}"
    config.save.path += f"/{config.save.name}"
    prompts = prompt_util.load_prompts_from_yaml(config.prompts_file, attributes)
    device = torch.device(f"cuda:{args.device}")
    folders = [f.strip() for f in args.folders.split(",")]
    scales = [float(s.strip()) for s in args.scales.split(",")]
    train(config, prompts, device, args.folder_main, folders, scales, rank=args.rank, xl=False)


if __name__ == "__main__":
    p = build_parser()
    p.set_defaults(config_file="data/config.yaml")
    main(p.parse_args())
