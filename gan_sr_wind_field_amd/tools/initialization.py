"""Weight initialisation (reference ``tools/initialization.py:15-34``).

Conv / Linear weights: ``kaiming_normal_(a=0, fan_in)`` then ``* scale``; their
biases zero.  Two reference behaviours are kept on purpose so that a seeded run
starts from the reference's exact weights (same RNG draws, same
``Module.apply`` order):

* BatchNorm layers stay at their constructor values - the reference's
  BatchNorm branch never matches (it tests for the string "BatchNorm3D").
* The growth convs inside ``RDB_Conv`` stay at ``nn.Conv3d``'s DEFAULT init: the
  reference wraps every ``RDB_Conv`` in ``torch.jit.script``
  (torch_blocks.py:256-267), its init matches on ``__class__.__name__``, and a
  scripted sub-module's class name is ``RecursiveScriptModule`` - so those 192
  convs are never re-initialised (and consume no RNG draws here).
"""
import torch.nn as nn
from torch.nn import init


def init_kaiming(m: nn.Module, scale: float = 1) -> None:
    if isinstance(m, (nn.Conv2d, nn.Conv3d, nn.Linear)):
        init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
        m.weight.data *= scale
        if m.bias is not None:
            m.bias.data.zero_()


def init_weights(m: nn.Module, scale: float = 1) -> None:
    from ..CNN_models.torch_blocks import RDB_Conv

    skipped = set()
    for mod in m.modules():
        if isinstance(mod, RDB_Conv):
            skipped.update(id(sub) for sub in mod.modules())

    def visit(mod: nn.Module) -> None:
        if id(mod) not in skipped:
            init_kaiming(mod, scale=scale)

    m.apply(visit)
