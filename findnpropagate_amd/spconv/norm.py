"""BatchNorm1d (training mode) + ReLU (+ residual) on the rows of a sparse tensor as ONE autograd node backed by
csrc/bnorm.hip (fnp_bn_train_forward / fnp_bn_train_backward).

The reference applies `nn.BatchNorm1d`, `nn.ReLU` and the residual add of SparseBasicBlock as separate torch modules on
`.features` (pcdet/models/backbones_3d/spconv_backbone.py:8-27,51-67); with 16-bit activations that is a cast to f32, the
normalisation, the activation and a cast back per layer.  `bn_act` does the same arithmetic (f32 math on the stored
values, biased batch variance, torch's running-statistics update) in two passes per direction, deterministically."""
import torch
import torch.nn as nn

from .. import lib as _l


class _BNActFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, n_dev, momentum, eps, relu, batches_tracked=None):
        L = _l.load()
        _l.require_device(x, n_dev)
        assert x.is_contiguous() and x.dim() == 2
        cap, C = x.shape
        y = torch.empty_like(x)
        mean = torch.empty((C,), dtype=torch.float32, device=x.device)
        invstd = torch.empty((C,), dtype=torch.float32, device=x.device)
        ws = torch.empty((int(L.fnp_bn_workspace_bytes(C)),), dtype=torch.uint8, device=x.device)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        if residual is not None:
            assert residual.shape == x.shape and residual.dtype == x.dtype and residual.is_contiguous()
        rc = L.fnp_bn_train_forward(_l.ptr(x), _l.dtype_code(x), _l.ptr(n_dev), max(cap, 1), C, _l.ptr(g32), _l.ptr(b32),
                                    _l.ptr(running_mean), _l.ptr(running_var), float(momentum), float(eps), _l.ptr(residual),
                                    int(bool(relu)), _l.ptr(y), _l.ptr(mean), _l.ptr(invstd), _l.ptr(batches_tracked), _l.ptr(ws), ws.numel(),
                                    _l.stream())
        _l.check(rc, "fnp_bn_train_forward")
        if batches_tracked is not None:
            torch.autograd.graph.increment_version(batches_tracked)
        # the kernel wrote the running statistics through raw pointers: tell torch (FusedResBackbone.prepare() keys its
        # folded BatchNorm constants on the buffers' versions)
        torch.autograd.graph.increment_version(running_mean)
        torch.autograd.graph.increment_version(running_var)
        ctx.save_for_backward(x, y, g32, mean, invstd, n_dev)
        ctx.relu, ctx.has_res, ctx.param_dtype = bool(relu), residual is not None, gamma.dtype
        return y

    @staticmethod
    def backward(ctx, grad_out):
        L = _l.load()
        x, y, g32, mean, invstd, n_dev = ctx.saved_tensors
        grad_out = grad_out.contiguous().to(x.dtype)
        cap, C = x.shape
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        dgamma = torch.empty((C,), dtype=torch.float32, device=x.device)
        dbeta = torch.empty((C,), dtype=torch.float32, device=x.device)
        ws = torch.empty((int(L.fnp_bn_workspace_bytes(C)),), dtype=torch.uint8, device=x.device)
        rc = L.fnp_bn_train_backward(_l.ptr(grad_out), _l.ptr(x), _l.ptr(y), _l.dtype_code(x), _l.ptr(n_dev), max(cap, 1), C,
                                     _l.ptr(g32), _l.ptr(mean), _l.ptr(invstd), int(ctx.relu), _l.ptr(dx), _l.ptr(dres),
                                     _l.ptr(dgamma), _l.ptr(dbeta), _l.ptr(ws), ws.numel(), _l.stream())
        _l.check(rc, "fnp_bn_train_backward")
        return dx, dgamma.to(ctx.param_dtype), dbeta.to(ctx.param_dtype), dres, None, None, None, None, None, None, None


ENABLED = True     # (tests switch it off to compare with the unfused torch modules)


def fusable(bn):
    """a BatchNorm1d in training mode with affine parameters and a fixed momentum, on a channel count the kernels take"""
    return (ENABLED and isinstance(bn, nn.BatchNorm1d) and bn.training and bn.affine and bn.momentum is not None and bn.track_running_stats
            and bn.num_features in (8, 16, 32, 64, 128, 256))


def bn_act(x, n_dev, bn, residual=None, relu=True):
    """y = relu(bn(x) [+ residual]) over the first n_dev rows of x (cap, C) in x's dtype; bn: nn.BatchNorm1d in training
    mode (its running statistics and num_batches_tracked advance exactly like torch's forward)."""
    assert fusable(bn)
    nbt = bn.num_batches_tracked
    if not (nbt.is_cuda and nbt.dtype == torch.int64):
        with torch.no_grad():
            bn.num_batches_tracked += 1
        nbt = None
    return _BNActFunction.apply(x.contiguous(), bn.weight, bn.bias, None if residual is None else residual.contiguous(),
                                bn.running_mean, bn.running_var, n_dev, bn.momentum, bn.eps, relu, nbt)
