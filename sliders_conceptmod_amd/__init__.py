"""MI355X-native concept-slider LoRA trainer: host-side mirror of ntc-ai/sliders-conceptmod's trainer modules over the
C ABI in include/smi.h (libsmi_hip.so, hand-written HIP for gfx950).  Nothing is imported eagerly: `import
sliders_conceptmod_amd.lora` etc. pull in PyTorch; the HIP library is loaded on first use and its absence is an error."""
__all__ = ["lora", "unet", "train_util", "prompt_util", "config_util", "model_util", "step", "parallel",
           "train_lora", "train_lora_xl", "train_lora_scale_xl", "build"]
