"""nerfstudio plugin surface: what `ns-train` / `ns-eval` / `ns-eval-unc` find when this package is installed
next to nerfstudio.

The reference registers four methods under the entry-point group `nerfstudio.method_configs`
(/root/reference/pyproject.toml:18-22); `pyproject.toml` of this repo lists the same four names, each resolving
to an attribute of this module:

    dropout          -> NerfactoMCDropoutMethod   method_name "nerfacto-mcdropout"  (mcdropout_configs.py:17-54)
    laplace_d        -> NerfactoLaplaceMethod     method_name "nerfacto-laplace"    (laplace_config.py:21-58)
    activenerfacto   -> ActiveNerfactoMethod      method_name "active-nerfacto"     (activenerfacto_config.py:24-61)
    activesplatfacto -> ActiveSplatfactoMethod    method_name "active-splatfacto"   (activesplatfacto_config.py:32-92)

Each is a `MethodSpecification(TrainerConfig(method_name=..., pipeline=VanillaPipelineConfig(datamanager=...,
model=<ModelConfig of this build>), optimizers=..., viewer=...), description=...)` with the reference's values, and
each ModelConfig's `_target` is a `nerfstudio.models.base_model.Model` subclass defined here whose rendering methods
forward to the HIP-kernel mirrors of `models.py`:

    get_outputs(ray_bundle: RayBundle)                      one chunk of rays
    get_outputs_for_camera_ray_bundle(camera_ray_bundle)    [H,W] bundle (mcdropout_models.py:94-96)
    get_outputs_for_camera(camera, obb_box=None)            (eval_uncertainty.py:1097)
    NerfactoLaplaceModel.get_outputs_for_camera_unc(...)    (laplace_model.py:403-415), compute_hessian_naive (:343)
    load_state_dict / get_param_groups / get_training_callbacks / get_metrics_dict / get_image_metrics_and_images

nerfstudio is not installed in the build image, so the nerfstudio-dependent objects are built lazily (module
`__getattr__`, PEP 562): importing this module never needs nerfstudio, resolving one of the four entry-point
attributes does.  `METHOD_NAMES`, `MODEL_CONFIGS` and `build_model` work without it (the stand-alone eval harness
uses them).  The scope is inference: `get_loss_dict` raises -- training losses are not part of this build.
"""
from __future__ import annotations

from typing import Any, Dict, List, Tuple

import torch

from . import models

METHOD_NAMES = ("nerfacto-mcdropout", "nerfacto-laplace", "active-nerfacto", "active-splatfacto")

# entry-point name (pyproject.toml) -> (attribute of this module, method name)
ENTRY_POINTS = {
    "dropout": ("NerfactoMCDropoutMethod", "nerfacto-mcdropout"),
    "laplace_d": ("NerfactoLaplaceMethod", "nerfacto-laplace"),
    "activenerfacto": ("ActiveNerfactoMethod", "active-nerfacto"),
    "activesplatfacto": ("ActiveSplatfactoMethod", "active-splatfacto"),
}

MODEL_CONFIGS = {
    # the reference's method configs set eval_num_rays_per_chunk = 1<<15 and average_init_density = 0.01
    # (mcdropout_configs.py:31-32, laplace_config.py:35-36, activenerfacto_config.py:38-39)
    "nerfacto-mcdropout": lambda: models.NerfactoMCDropoutModelConfig(eval_num_rays_per_chunk=1 << 15,
                                                                      average_init_density=0.01),
    "nerfacto-laplace": lambda: models.NerfactoLaplaceModelConfig(eval_num_rays_per_chunk=1 << 15,
                                                                  average_init_density=0.01),
    "active-nerfacto": lambda: models.ActiveNerfactoModelConfig(eval_num_rays_per_chunk=1 << 15,
                                                                average_init_density=0.01),
    "active-splatfacto": lambda: models.ActiveSplatfactoModelConfig(),
    # upstream's own methods, as ensemble members (README.md:106-108: "train a nerfacto or splatfacto model using
    # different seeds"; ensemble_utils.py:149-156).  nerfstudio registers these two itself: no entry point here.
    # [UPSTREAM nerfstudio 1.1.0 method_configs["nerfacto"]: eval_num_rays_per_chunk = 1 << 15, average_init_density = 0.01]
    "nerfacto": lambda: models.PlainNerfactoModelConfig(eval_num_rays_per_chunk=1 << 15, average_init_density=0.01),
    "splatfacto": lambda: models.SplatfactoModelConfig(),
}

DESCRIPTIONS = {   # the reference's description strings
    "nerfacto-mcdropout": "Nerfacto with dropout for RGB and density",
    "nerfacto-laplace": "LaplaceNerf for Nerfacto model. By default uses the basic configurations of Nerfacto.",
    "active-nerfacto": "Nerfacto-variant of ActiveNerf with predicted uncertainty for RGB",
    "active-splatfacto": "Splatfacto-variant of ActiveNerf with rendered uncertainty for RGB",
}

_MIRRORS = {
    "nerfacto-mcdropout": (models.NerfactoMCDropoutModel, models.NerfactoMCDropoutModelConfig),
    "nerfacto-laplace": (models.NerfactoLaplaceModel, models.NerfactoLaplaceModelConfig),
    "active-nerfacto": (models.ActiveNerfactoModel, models.ActiveNerfactoModelConfig),
    "active-splatfacto": (models.ActiveSplatfactoModel, models.ActiveSplatfactoModelConfig),
}


def build_model(method_name: str, **kw):
    """Instantiate the eval-side model of a method with the reference's config values (no nerfstudio needed)."""
    cfg = MODEL_CONFIGS[method_name]()
    return cfg._target(cfg, **kw)


# --------------------------------------------------------------------------------------------------------------
# nerfstudio-facing classes, built on first use
# --------------------------------------------------------------------------------------------------------------
_BUILT: Dict[str, Any] = {}


def _require_nerfstudio():
    try:
        import nerfstudio  # noqa: F401
    except ImportError as e:
        raise ImportError("nerfstudio is required to register the methods with ns-train / ns-eval "
                          "(uncertainty_nerf_gs_amd.plugin.build_model works without it)") from e


def _build() -> Dict[str, Any]:
    """Define the nerfstudio Model / ModelConfig subclasses and the four MethodSpecifications."""
    if _BUILT:
        return _BUILT
    _require_nerfstudio()
    import dataclasses
    from dataclasses import dataclass, field

    from nerfstudio.cameras.camera_optimizers import CameraOptimizerConfig
    from nerfstudio.configs.base_config import ViewerConfig
    from nerfstudio.data.datamanagers.base_datamanager import VanillaDataManagerConfig
    from nerfstudio.data.datamanagers.full_images_datamanager import FullImageDatamanagerConfig
    from nerfstudio.data.dataparsers.nerfstudio_dataparser import NerfstudioDataParserConfig
    from nerfstudio.engine.optimizers import AdamOptimizerConfig
    from nerfstudio.engine.schedulers import ExponentialDecaySchedulerConfig
    from nerfstudio.engine.trainer import TrainerConfig
    from nerfstudio.models.base_model import Model, ModelConfig
    from nerfstudio.pipelines.base_pipeline import VanillaPipelineConfig
    from nerfstudio.plugins.types import MethodSpecification

    # ---- Model subclasses: nerfstudio owns construction / checkpoints / devices, the mirror renders --------------
    class _HipModel(Model):
        """Common part: the mirror of `models.py` is built in populate_modules and its submodules are registered
        under the reference's names (`field`, `proposal_networks` / `gauss_params`), so nerfstudio checkpoints of the
        reference load key for key."""
        method_name = ""

        def populate_modules(self):
            super().populate_modules()
            mirror_cls, mirror_cfg_cls = _MIRRORS[self.method_name]
            names = {f.name for f in dataclasses.fields(mirror_cfg_cls)} - {"_target"}
            mcfg = mirror_cfg_cls(**{k: getattr(self.config, k) for k in names if hasattr(self.config, k)})
            kw = dict(self.kwargs) if isinstance(getattr(self, "kwargs", None), dict) else {}
            mirror = self._make_mirror(mirror_cls, mcfg, kw)
            object.__setattr__(self, "_mirror", mirror)      # not a registered child: parameters are shared below
            self._adopt(mirror)

        def _make_mirror(self, mirror_cls, mcfg, kw):
            return mirror_cls(mcfg, scene_box=self.scene_box, num_train_data=self.num_train_data)

        def _adopt(self, mirror):
            self.field = mirror.field
            self.proposal_networks = mirror.proposal_networks

        # -- training-side hooks nerfstudio calls while setting a pipeline up ---------------------------------------
        def get_param_groups(self) -> Dict[str, List[torch.nn.Parameter]]:
            return {"proposal_networks": list(self.proposal_networks.parameters()), "fields": list(self.field.parameters())}

        def get_training_callbacks(self, training_callback_attributes) -> List:
            return []

        def get_loss_dict(self, outputs, batch, metrics_dict=None):
            raise NotImplementedError("this build is the inference / evaluation path of the method; train with the "
                                      "reference and load the checkpoint here")

        # -- rendering ---------------------------------------------------------------------------------------------
        def get_outputs(self, ray_bundle):
            return self._mirror.get_outputs(ray_bundle)

        def forward(self, ray_bundle):
            return self._mirror.forward(ray_bundle)

        @torch.no_grad()
        def get_outputs_for_camera_ray_bundle(self, camera_ray_bundle):
            return self._mirror.get_outputs_for_camera_ray_bundle(camera_ray_bundle)

        @torch.no_grad()
        def get_outputs_for_camera(self, camera, obb_box=None):
            return self._mirror.get_outputs_for_camera(camera, obb_box=obb_box)

        def load_state_dict(self, state_dict, strict: bool = False, **kw):  # type: ignore[override]
            return self._mirror.load_state_dict(state_dict, strict=strict, **kw)

        # -- metrics (eval_uncertainty.py:683-684 calls model.psnr / model.ssim) -----------------------------------
        def psnr(self, image, rgb):
            return self._mirror.psnr(image, rgb)

        def ssim(self, image, rgb):
            return self._mirror.ssim(image, rgb)

        def get_metrics_dict(self, outputs, batch) -> Dict[str, torch.Tensor]:
            gt = batch["image"].to(outputs["rgb"].device)
            return {"psnr": self.psnr(torch.moveaxis(gt, -1, 0)[None, ...], torch.moveaxis(outputs["rgb"], -1, 0)[None, ...])}

        def get_image_metrics_and_images(self, outputs, batch) -> Tuple[Dict[str, float], Dict[str, torch.Tensor]]:
            gt = batch["image"].to(outputs["rgb"].device)
            a, b = torch.moveaxis(gt, -1, 0)[None, ...], torch.moveaxis(outputs["rgb"], -1, 0)[None, ...]
            metrics = {"psnr": float(self.psnr(a, b)), "ssim": float(self.ssim(a, b))}
            return metrics, {"img": torch.cat([gt, outputs["rgb"]], dim=1)}

    class ActiveNerfactoModel(_HipModel):
        method_name = "active-nerfacto"

    class NerfactoMCDropoutModel(_HipModel):
        method_name = "nerfacto-mcdropout"

    class NerfactoLaplaceModel(_HipModel):
        method_name = "nerfacto-laplace"

        @torch.no_grad()
        def get_outputs_for_camera_unc(self, camera, obb_box=None, is_inference: bool = True,
                                       use_deterministic_density: bool = False, prior_prec: float = 1.0,
                                       n_samples: int = 100, eps: float = 1e-9):
            return self._mirror.get_outputs_for_camera_unc(camera, obb_box=obb_box, is_inference=is_inference,
                                                           use_deterministic_density=use_deterministic_density,
                                                           prior_prec=prior_prec, n_samples=n_samples, eps=eps)

        def compute_hessian_naive(self, pipeline=None, n_iters: int = 1000, **kw):
            return self._mirror.compute_hessian_naive(pipeline, n_iters=n_iters, **kw)

    class ActiveSplatfactoModel(_HipModel):
        method_name = "active-splatfacto"

        def _make_mirror(self, mirror_cls, mcfg, kw):
            seed = kw.get("seed_points")
            n = int(seed[0].shape[0]) if seed is not None else 1000
            m = mirror_cls(mcfg, num_points=n)
            if seed is not None:                     # [UPSTREAM SplatfactoModel.populate_modules] means from the SfM points
                with torch.no_grad():
                    m.gauss_params["means"].copy_(seed[0])
            return m

        def _adopt(self, mirror):
            self.gauss_params = mirror.gauss_params

        def get_param_groups(self):
            return {name: [self.gauss_params[name]] for name in models.ActiveSplatfactoModel.GAUSS}

        def load_state_dict(self, dict, **kwargs):  # type: ignore[override]
            out = self._mirror.load_state_dict(dict, **kwargs)
            self.gauss_params = self._mirror.gauss_params       # resized to the checkpoint's point count (:87-100)
            return out

        def get_outputs(self, camera):
            return self._mirror.get_outputs(camera)

        forward = get_outputs

        def set_crop(self, crop_box):
            self._mirror.set_crop(crop_box)

        def set_background(self, background_color):
            self._mirror.set_background(background_color)

        def get_gt_img(self, image):
            return self._mirror.get_gt_img(image)

        def composite_with_background(self, image, background):
            return self._mirror.composite_with_background(image, background)

    # ---- ModelConfigs: nerfstudio's InstantiateConfig machinery + the fields of this build's configs ------------
    def _ns_config(name: str, mirror_cfg_cls, target):
        """dataclass(ModelConfig) carrying every field of the mirror config (same names and defaults as the
        reference's config for the fields that shape eval rendering) plus `camera_optimizer` for the NeRF methods"""
        # models.py uses `from __future__ import annotations`: Field.type holds STRINGS there.  tyro / dataclasses resolve a
        # dynamic class's string annotations in the namespace of the module that CREATED the class (this one), where
        # they would only resolve by luck -- hand over the evaluated types instead
        import typing
        hints = typing.get_type_hints(mirror_cfg_cls)
        ann, ns = {}, {}
        for f in dataclasses.fields(mirror_cfg_cls):
            if f.name == "_target":
                continue
            ann[f.name] = hints[f.name]
            ns[f.name] = (field(default_factory=f.default_factory) if f.default_factory is not dataclasses.MISSING
                          else f.default)
        ann["_target"] = type
        ns["_target"] = field(default_factory=lambda: target)
        if name != "active-splatfacto":
            ann["camera_optimizer"] = Any
            ns["camera_optimizer"] = field(default_factory=lambda: CameraOptimizerConfig(mode="SO3xR3"))
        ns["__annotations__"] = ann
        ns["__doc__"] = f"ModelConfig of {name} (HIP-kernel build); fields follow the reference's config"
        return dataclass(type(target.__name__ + "Config", (ModelConfig,), ns))

    cfgs = {
        "active-nerfacto": _ns_config("active-nerfacto", models.ActiveNerfactoModelConfig, ActiveNerfactoModel),
        "nerfacto-mcdropout": _ns_config("nerfacto-mcdropout", models.NerfactoMCDropoutModelConfig, NerfactoMCDropoutModel),
        "nerfacto-laplace": _ns_config("nerfacto-laplace", models.NerfactoLaplaceModelConfig, NerfactoLaplaceModel),
        "active-splatfacto": _ns_config("active-splatfacto", models.ActiveSplatfactoModelConfig, ActiveSplatfactoModel),
    }

    # ---- MethodSpecifications with the reference's trainer values ------------------------------------------------
    def _nerf_spec(name: str):
        sched = lambda: ExponentialDecaySchedulerConfig(lr_final=0.0001, max_steps=200000)
        return MethodSpecification(
            TrainerConfig(
                method_name=name, steps_per_eval_batch=500, steps_per_save=2000, max_num_iterations=30000,
                mixed_precision=True,
                pipeline=VanillaPipelineConfig(
                    datamanager=VanillaDataManagerConfig(dataparser=NerfstudioDataParserConfig(),
                                                         train_num_rays_per_batch=4096, eval_num_rays_per_batch=4096),
                    model=cfgs[name](eval_num_rays_per_chunk=1 << 15, average_init_density=0.01,
                                     camera_optimizer=CameraOptimizerConfig(mode="SO3xR3"))),
                optimizers={
                    "proposal_networks": {"optimizer": AdamOptimizerConfig(lr=1e-2, eps=1e-15), "scheduler": sched()},
                    "fields": {"optimizer": AdamOptimizerConfig(lr=1e-2, eps=1e-15), "scheduler": sched()},
                    "camera_opt": {"optimizer": AdamOptimizerConfig(lr=1e-3, eps=1e-15),
                                   "scheduler": ExponentialDecaySchedulerConfig(lr_final=1e-4, max_steps=5000)},
                },
                viewer=ViewerConfig(num_rays_per_chunk=1 << 15), vis="viewer"),
            description=DESCRIPTIONS[name])

    def _splat_spec():
        adam = lambda lr: {"optimizer": AdamOptimizerConfig(lr=lr, eps=1e-15), "scheduler": None}
        return MethodSpecification(
            config=TrainerConfig(
                method_name="active-splatfacto", steps_per_eval_image=100, steps_per_eval_batch=0, steps_per_save=2000,
                steps_per_eval_all_images=1000, max_num_iterations=30000, mixed_precision=False,
                pipeline=VanillaPipelineConfig(
                    datamanager=FullImageDatamanagerConfig(dataparser=NerfstudioDataParserConfig(load_3D_points=True),
                                                           cache_images_type="uint8"),
                    model=cfgs["active-splatfacto"]()),
                optimizers={
                    "means": {"optimizer": AdamOptimizerConfig(lr=1.6e-4, eps=1e-15),
                              "scheduler": ExponentialDecaySchedulerConfig(lr_final=1.6e-6, max_steps=30000)},
                    "features_dc": adam(0.0025), "features_rest": adam(0.0025 / 20), "opacities": adam(0.05),
                    "scales": adam(0.005), "quats": adam(0.001), "log_uncertainties": adam(0.0025),
                    "camera_opt": {"optimizer": AdamOptimizerConfig(lr=1e-4, eps=1e-15),
                                   "scheduler": ExponentialDecaySchedulerConfig(lr_final=5e-7, max_steps=30000,
                                                                                warmup_steps=1000, lr_pre_warmup=0)},
                },
                viewer=ViewerConfig(num_rays_per_chunk=1 << 15), vis="viewer"),
            description=DESCRIPTIONS["active-splatfacto"])

    _BUILT.update({
        "ActiveNerfactoModel": ActiveNerfactoModel, "NerfactoMCDropoutModel": NerfactoMCDropoutModel,
        "NerfactoLaplaceModel": NerfactoLaplaceModel, "ActiveSplatfactoModel": ActiveSplatfactoModel,
        "ActiveNerfactoModelConfig": cfgs["active-nerfacto"], "NerfactoMCDropoutModelConfig": cfgs["nerfacto-mcdropout"],
        "NerfactoLaplaceModelConfig": cfgs["nerfacto-laplace"], "ActiveSplatfactoModelConfig": cfgs["active-splatfacto"],
        "NerfactoMCDropoutMethod": _nerf_spec("nerfacto-mcdropout"), "NerfactoLaplaceMethod": _nerf_spec("nerfacto-laplace"),
        "ActiveNerfactoMethod": _nerf_spec("active-nerfacto"), "ActiveSplatfactoMethod": _splat_spec(),
    })
    return _BUILT


_LAZY = ("ActiveNerfactoModel", "NerfactoMCDropoutModel", "NerfactoLaplaceModel", "ActiveSplatfactoModel",
         "ActiveNerfactoModelConfig", "NerfactoMCDropoutModelConfig", "NerfactoLaplaceModelConfig",
         "ActiveSplatfactoModelConfig", "NerfactoMCDropoutMethod", "NerfactoLaplaceMethod", "ActiveNerfactoMethod",
         "ActiveSplatfactoMethod")


def __getattr__(name: str):
    if name in _LAZY:
        return _build()[name]
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def method_specifications() -> Dict[str, object]:
    """method name -> MethodSpecification (needs nerfstudio installed); what the entry points resolve to"""
    built = _build()
    return {method: built[attr] for attr, method in ENTRY_POINTS.values()}
