"""Import the REAL reference (``/root/reference``) for oracle validation / golden generation.

TEST INFRASTRUCTURE ONLY.  Works only in the build container (the GPU box has no
``/root/reference``); nothing in ``-m gpu`` tests, ``smoke()`` or ``bench.py`` may call this.

``models/latte.py:16`` imports ``timm.models.vision_transformer.{Mlp,PatchEmbed}``; timm is not
installed here, so a stand-in module with the same attribute names / forward semantics is
injected into ``sys.modules`` and the reference file is loaded *by path*, unmodified.
(``import models`` is avoided: ``models/__init__.py:6-7`` pulls diffusers.)
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("LATTE_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "latte.py"))


class _Mlp(nn.Module):
    """timm ``Mlp``: fc1 -> act -> drop1 -> norm(Identity) -> fc2 -> drop2 (used at latte.py:171)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU,
                 norm_layer=None, bias=True, drop=0.0, use_conv=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


class _PatchEmbed(nn.Module):
    """timm ``PatchEmbed``: Conv2d(k=s=patch) -> flatten(2).transpose(1,2) (used at latte.py:233)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None,
                 flatten=True, bias=True):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)
        self.norm = nn.Identity()

    def forward(self, x):
        x = self.proj(x)
        if self.flatten:
            x = x.flatten(2).transpose(1, 2)
        return self.norm(x)


def _install_timm_standin():
    if "timm.models.vision_transformer" in sys.modules:
        return
    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    vt = types.ModuleType("timm.models.vision_transformer")
    vt.Mlp = _Mlp
    vt.PatchEmbed = _PatchEmbed
    timm.models = timm_models
    timm_models.vision_transformer = vt
    sys.modules["timm"] = timm
    sys.modules["timm.models"] = timm_models
    sys.modules["timm.models.vision_transformer"] = vt


def load_reference_latte():
    """Returns the reference ``models/latte.py`` module object (unmodified source)."""
    assert reference_available(), "reference checkout not present (expected only in the build container)"
    _install_timm_standin()
    name = "_reference_latte"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE_ROOT, "models", "latte.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_diffusion():
    """Returns the reference ``diffusion`` package (numpy + torch only)."""
    assert reference_available()
    name = "_reference_diffusion"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(REFERENCE_ROOT, "diffusion", "__init__.py"),
        submodule_search_locations=[os.path.join(REFERENCE_ROOT, "diffusion")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def randomize_zero_init(model, std=0.02, seed=1234):
    """The reference zero-inits every adaLN gate and the final layer (latte.py:286-295), which makes
    the model output identically 0 and any parity test vacuous; re-draw those tensors N(0, std)."""
    g = torch.Generator("cpu").manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.requires_grad and float(p.abs().max()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * std)
    return model


def load_reference_latte_t2v():
    """Returns the reference ``models/latte_t2v.py`` module object (unmodified source) on top of the diffusers stand-in
    (oracle/diffusers_standin.py: memory-derived restatement of the diffusers 0.24.0 leaf modules it imports)."""
    assert reference_available(), "reference checkout not present (expected only in the build container)"
    from oracle import diffusers_standin
    diffusers_standin.install()
    name = "_reference_latte_t2v"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE_ROOT, "models", "latte_t2v.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod
