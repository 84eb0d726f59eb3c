"""Import harness for the *reference* (Bai-YT/ConsistencyTTA) numerical modules.

Only usable in the build container, where the read-only reference tree is mounted at
/root/reference.  It is used by the `make_golden*.py` scripts to produce the committed fixtures
in this directory; the CPU oracle is then checked against those fixtures
(`tests/test_oracle_golden.py`).  Nothing on the GPU box imports this file's targets:
/root/reference does not exist there.

Recipe follows SURVEY.md §8(c): the trimmed `easy_inference` copy of diffusers/audioldm is
arithmetically identical to the full vendored tree (diff = import lines) and imports under
the installed transformers/huggingface_hub once two hub symbols are stubbed.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("CTTA_REFERENCE_ROOT", "/root/reference")
_EASY = os.path.join(REF_ROOT, "easy_inference")


def available() -> bool:
    return os.path.isdir(_EASY)


_loaded = {}


def load():
    """Returns a namespace with the reference classes used as parity anchors."""
    if _loaded:
        return _loaded["ns"]
    if not available():
        raise RuntimeError("reference tree not mounted at %s" % REF_ROOT)
    sys.dont_write_bytecode = True  # /root/reference is read-only

    import huggingface_hub
    import huggingface_hub.constants as hc

    if not hasattr(huggingface_hub, "HfFolder"):
        class HfFolder:  # vendored diffusers expects the hub<=0.15 API
            @staticmethod
            def get_token():
                return None
        huggingface_hub.HfFolder = HfFolder
    if not hasattr(hc, "hf_cache_home"):
        hc.hf_cache_home = os.path.expanduser("~/.cache/huggingface")
    if not hasattr(huggingface_hub, "cached_download"):
        huggingface_hub.cached_download = lambda *a, **k: (_ for _ in ()).throw(
            RuntimeError("no network"))

    for name in ("wandb", "soundfile", "librosa", "resampy", "torchaudio", "laion_clap"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = types.ModuleType(name)
                m.__spec__ = types.SimpleNamespace(name=name, loader=None, origin=None,
                                                   submodule_search_locations=None)
                sys.modules[name] = m

    sys.path.insert(0, _EASY)
    try:
        import diffusers as ref_diffusers  # noqa: F401  (the trimmed easy_inference copy)
        from diffusers.models.unet_2d_condition_guided import UNet2DConditionGuidedModel
        from diffusers.models.unet_2d_condition import UNet2DConditionModel
        from diffusers.scheduling_heun_discrete import HeunDiscreteScheduler
        from diffusers.models import resnet as ref_resnet
        from diffusers.models import attention as ref_attention
        from diffusers.models import transformer_2d as ref_transformer_2d
        from diffusers.models import embeddings as ref_embeddings
        import audioldm.variational_autoencoder.autoencoder as ref_autoencoder
        import audioldm.variational_autoencoder.modules as ref_vae_modules
        import audioldm.hifigan.models as ref_hifigan_models
        import audioldm.hifigan.utilities as ref_hifigan_utilities
        from audioldm.utils import default_audioldm_config
    finally:
        pass

    ns = types.SimpleNamespace(
        UNet2DConditionGuidedModel=UNet2DConditionGuidedModel,
        UNet2DConditionModel=UNet2DConditionModel,
        HeunDiscreteScheduler=HeunDiscreteScheduler,
        resnet=ref_resnet,
        attention=ref_attention,
        transformer_2d=ref_transformer_2d,
        embeddings=ref_embeddings,
        autoencoder=ref_autoencoder,
        vae_modules=ref_vae_modules,
        hifigan_models=ref_hifigan_models,
        hifigan_utilities=ref_hifigan_utilities,
        default_audioldm_config=default_audioldm_config,
        light_config_path=os.path.join(REF_ROOT, "configs", "tango_diffusion_light.json"),
        full_config_path=os.path.join(REF_ROOT, "configs", "tango_diffusion.json"),
    )
    _loaded["ns"] = ns
    return ns


def make_heun(ns):
    """SD-2.1 scheduler config (fetched from the HF hub by the reference, train.sh:5);
    values restated in SURVEY.md §2a."""
    return ns.HeunDiscreteScheduler(
        num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
        beta_schedule="scaled_linear", prediction_type="v_prediction")
