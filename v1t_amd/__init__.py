"""v1t_amd — MI355X-native (gfx950) V1T hot path: ViT core + Gaussian2d readout behind the reference's
core / readout plugin registries. See DESIGN.md and INTEGRATION.md."""
from .core import Core, ViTCore, get_core, register as register_core  # noqa: F401
from .cct import CCTCore  # noqa: F401  (importing it fills the "cct" registry entry, as core/__init__.py:1 does in the reference)
from .readout import Gaussian2DReadout, Readout, Readouts, register as register_readout  # noqa: F401
from .model import ELU1, CoreShifter, CoreShifters, ImageCropper, Model  # noqa: F401
from .trainer import FusedAdamW  # noqa: F401  (opt-in optimizer for the reference's own loop: FusedAdamW.for_model)

__all__ = ["Core", "ViTCore", "CCTCore", "get_core", "register_core", "Gaussian2DReadout", "Readout", "Readouts", "register_readout",
           "ELU1", "CoreShifter", "CoreShifters", "ImageCropper", "Model", "FusedAdamW", "install_into_reference"]


def install_into_reference() -> bool:
    """If the reference package `v1t` is importable, overwrite its registry entries "vit", "cct" and
    "gaussian2d" (core/core.py:8-16, readout/readout.py:10-18) with the native classes - and the "poisson" entry of its criterion
    registry (losses.py:9-17) with the fused PoissonLoss - so that the reference's own `train.py --core vit --readout gaussian2d`
    (or `--core cct`) picks them up unchanged."""
    try:
        from v1t.models.core import core as ref_core  # type: ignore
        from v1t.models.readout import readout as ref_readout  # type: ignore
    except Exception:
        return False
    ref_core.register("vit")(ViTCore)
    ref_core.register("cct")(CCTCore)
    ref_readout.register("gaussian2d")(Gaussian2DReadout)
    try:  # the criterion registry (losses.py:9-17, 141-166): the same PoissonLoss as one fused launch on fp32 GPU tensors
        from v1t import losses as ref_losses  # type: ignore

        from .losses import PoissonLoss

        ref_losses.register("poisson")(PoissonLoss)
    except Exception:
        pass
    try:  # the reference's Recorder looks for instances of ITS Attention class and hooks their `.attend` (utils/attention_rollout.py:24-36)
        from v1t.models.core import vit as ref_vit  # type: ignore

        ViTCore._reference_attention_cls = ref_vit.Attention  # cores built from now on carry one parameter-free tap per block
    except Exception:
        pass
    return True
