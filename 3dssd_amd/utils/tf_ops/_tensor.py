"""Small helpers shared by the operator wrappers: argument checks in the style of the reference's
OP_REQUIRES (raised as ValueError), and device/dtype plumbing on torch tensors."""
import torch


def require(cond, msg):
    if not cond:
        raise ValueError(msg)


def f32_cuda(t, name):
    require(isinstance(t, torch.Tensor), "%s must be a torch.Tensor" % name)
    require(t.is_cuda, "%s must live on the GPU (the HIP path has no CPU fallback)" % name)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def i32_cuda(t, name):
    require(isinstance(t, torch.Tensor), "%s must be a torch.Tensor" % name)
    require(t.is_cuda, "%s must live on the GPU (the HIP path has no CPU fallback)" % name)
    if t.dtype != torch.int32:
        t = t.int()
    return t.contiguous()
