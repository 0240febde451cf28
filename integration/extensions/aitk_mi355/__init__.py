"""ai-toolkit extension: the MI355X-native denoisers as model plug-ins.

Install: copy (or symlink) this directory to `<ai-toolkit>/extensions/aitk_mi355/` and put this repository on PYTHONPATH.  ai-toolkit
collects `AI_TOOLKIT_MODELS` from every package under `extensions/` and `extensions_built_in/` and selects a class by
`model.arch` (toolkit/util/get_model.py:20-50), so a config with `model: {arch: flux_mi355, name_or_path: <FLUX.1-dev diffusers dir>}`
(or `wan21_mi355`, `sd_mi355`) trains through the HIP kernels with the reference's own trainer, network and data code.

Each entry is a REAL `toolkit.models.base_model.BaseModel` subclass: the mirror from ai_toolkit_amd.plugin comes first in the MRO, so its
hooks (`load_model`, `get_noise_prediction`, `get_loss_target`, `encode_images`, `save_model`, ...) override BaseModel's, and everything the
mirror does not define (device-state presets, `prepare_optimizer_params`, hooks, ...) is BaseModel's own code.
"""
from toolkit.models.base_model import BaseModel

import ai_toolkit_amd  # noqa: F401  (import alias of the hyphenated package directory)
from ai_toolkit_amd import plugin as _p


def _flowmatch_scheduler(config):
    """The reference's OWN training scheduler, like every in-tree flow-matching plug-in returns from get_train_scheduler (e.g.
    extensions_built_in/diffusion_models/flux_kontext/flux_kontext.py:28-36, 412-414): the trainer's process_general_training_batch talks to it
    through the diffusers scheduler API (set_train_timesteps / timesteps / add_noise / config / get_weights_for_timesteps,
    jobs/process/BaseSDTrainProcess.py:1188-1323) — plumbing on 1000-entry tables, not part of the accelerated path."""
    def get_train_scheduler():
        from toolkit.samplers.custom_flowmatch_sampler import CustomFlowMatchEulerDiscreteScheduler

        return CustomFlowMatchEulerDiscreteScheduler(**config)

    return staticmethod(get_train_scheduler)


def _ddpm_scheduler():
    def get_train_scheduler():
        from toolkit.sampler import get_sampler  # the legacy StableDiffusion path builds its DDPM scheduler here (BaseSDTrainProcess.py:1773-1786)

        return get_sampler("ddpm", {"prediction_type": "epsilon"}, arch="sd")

    return staticmethod(get_train_scheduler)


def _real(mirror, get_train_scheduler):
    def __init__(self, device, model_config, dtype="bf16", custom_pipeline=None, noise_scheduler=None, **kwargs):
        BaseModel.__init__(self, device, model_config, dtype=dtype, custom_pipeline=custom_pipeline, noise_scheduler=noise_scheduler, **kwargs)
        mirror.__init__(self, device, model_config, dtype, custom_pipeline, noise_scheduler, **kwargs)

    return type(mirror.__name__.replace("Model", ""), (mirror, BaseModel),
                {"__init__": __init__, "__doc__": mirror.__doc__, "arch": mirror.arch, "get_train_scheduler": get_train_scheduler})


Flux1MI355 = _real(_p.Flux1MI355Model, _flowmatch_scheduler(_p.FLUX_SCHEDULER_CONFIG))
Wan21MI355 = _real(_p.Wan21MI355Model, _flowmatch_scheduler(_p.WAN_SCHEDULER_CONFIG))
StableDiffusionMI355 = _real(_p.StableDiffusionMI355Model, _ddpm_scheduler())

AI_TOOLKIT_MODELS = [Flux1MI355, Wan21MI355, StableDiffusionMI355]

# The trainer's optimizer tail on the arena kernels (ai_toolkit_amd/adopt.py, "The trainer's optimizer tail on the arena kernels"): once a network
# is adopted, a plain torch.optim.AdamW over exactly its parameters is stepped by ONE aitk_adamw_ema_step (torch optimizer step hooks, installed at
# adoption) and `ExponentialMovingAverage.update()` over the same parameters by ONE aitk_ema_update instead of its per-parameter Python loop
# (toolkit/ema.py:126-152) — wrapped here, where the reference's class is importable; everything else falls through to the reference's own code.
try:
    from toolkit.ema import ExponentialMovingAverage as _EMA

    from ai_toolkit_amd.adopt import install_ema_fusion as _install_ema_fusion

    _install_ema_fusion(_EMA)
except ImportError:  # a reference tree without toolkit/ema.py: nothing to wrap
    pass
