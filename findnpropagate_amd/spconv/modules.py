"""SparseModule / SparseSequential (spconv_backbone.py:21-25,30,193-234 use them)."""
from collections import OrderedDict

import torch.nn as nn

from .core import SparseConvTensor


class SparseModule(nn.Module):
    """Marker base class: modules that take and return a SparseConvTensor."""


def is_spconv_module(module):
    return isinstance(module, SparseModule)


class SparseSequential(SparseModule):
    """Sequential container: sparse modules get the tensor, dense nn.Modules get `.features`."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError(f"index {idx} is out of range")
        if idx < 0:
            idx += len(self)
        it = iter(self._modules.values())
        for _ in range(idx):
            next(it)
        return next(it)

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        for module in self._modules.values():
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.features.shape[0] > 0:
                    input = input.replace_feature(module(input.features))
            else:
                input = module(input)
        return input
