"""Small helpers shared by the operator wrappers: argument checks in the style of the reference's
OP_REQUIRES (raised as ValueError), and device/dtype plumbing on torch tensors."""
import torch


def require(cond, msg):
    if not cond:
        raise ValueError(msg)


def _on_current_device(t, name):
    # kernels are launched on the CURRENT device's stream (_native.current_stream): a tensor of another GPU would be
    # dereferenced on the wrong device
    require(t.device.index == torch.cuda.current_device(),
            "%s lives on cuda:%s but the current device is cuda:%d (use torch.cuda.device / set_device)"
            % (name, t.device.index, torch.cuda.current_device()))


def f32_cuda(t, name):
    require(isinstance(t, torch.Tensor), "%s must be a torch.Tensor" % name)
    require(t.is_cuda, "%s must live on the GPU (the HIP path has no CPU fallback)" % name)
    _on_current_device(t, name)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def i32_cuda(t, name):
    require(isinstance(t, torch.Tensor), "%s must be a torch.Tensor" % name)
    require(t.is_cuda, "%s must live on the GPU (the HIP path has no CPU fallback)" % name)
    _on_current_device(t, name)
    if t.dtype != torch.int32:
        t = t.int()
    return t.contiguous()
