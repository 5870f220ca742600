"""Shared host plumbing: parameter containers keyed like the reference's state dicts, the
per-device HIP context cache, workspace tensors and stream handles.  PyTorch is used for device
memory, streams and nn.Module bookkeeping only -- every FLOP runs in libvae_tagger_hip.so."""
import ctypes

import torch
import torch.nn as nn

from . import _lib, synth

_BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked")


class ParamTree(nn.Module):
    """nn.Module whose state_dict() keys are exactly `manifest`'s dotted names.

    Gives load_state_dict(strict=False) -> (missing, unexpected), .parameters(), .to(), .eval()
    -- the surface the reference relies on (diffusers_vae_loader.py:44, infer_full.py:27-28,63-67)
    -- without restating the reference's module classes.  Tensors not provided by a checkpoint keep
    the seeded default initialisation of vae_tagger_amd.synth."""

    def __init__(self, manifest, seed=0):
        super().__init__()
        self._manifest = dict(manifest)
        synth._note_fan_in(self._manifest)
        for key, shape in self._manifest.items():
            parts = key.split(".")
            node = self
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, nn.Module())
                node = node._modules[p]
            t = synth.synth_tensor(key, shape, seed)
            if parts[-1] in _BUFFER_LEAVES:
                node.register_buffer(parts[-1], t)
            else:
                node.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))


class HipModule(ParamTree):
    """ParamTree + lazily (re)uploaded packed weights inside a vt_context."""

    def __init__(self, manifest, seed=0):
        super().__init__(manifest, seed)
        self._ctx = None
        self._uploaded_version = None

    # any mutation path that matters for inference bumps the version so weights are re-packed
    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._uploaded_version = None
        return out

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._uploaded_version = None
        return out

    def _device_index(self):
        p = next(self.parameters())
        if p.device.type != "cuda":
            raise _lib.VTError(
                "vae_tagger_amd runs on MI355X only: move the model to a HIP device first "
                "(`.to('cuda')`). There is no CPU fallback.")
        return p.device.index if p.device.index is not None else torch.cuda.current_device()

    def _context(self):
        dev = self._device_index()
        if self._ctx is None or self._ctx.device_index != dev:
            self._ctx = _lib.Context(dev)
            self._uploaded_version = None
        if self._uploaded_version is None:
            self._upload(self._ctx)
            self._uploaded_version = 1
        return self._ctx

    def _upload(self, ctx):  # pragma: no cover - overridden
        raise NotImplementedError


_workspaces = {}


def workspace(device, nbytes, tag=""):
    """Grow-only per-device scratch tensor (caller-owned buffer of the C ABI); `tag` names an independent buffer."""
    key = (device.type, device.index, tag)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        _workspaces[key] = None
        ws = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    ptr = (ws.data_ptr() + 255) // 256 * 256
    return ws, ptr


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def vp(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def as_input(x):
    """fp32 NCHW contiguous device tensor (the dtype/layout the reference hands to vae.encode)."""
    if x.dim() != 4 or x.shape[1] != 3:
        raise ValueError(f"expected [B,3,H,W], got {tuple(x.shape)}")
    return x.detach().to(torch.float32).contiguous()
