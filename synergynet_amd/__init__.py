"""synergynet_amd: MI355X (gfx950) implementation of SynergyNet's inference hot path.

    from synergynet_amd import SynergyNet          # or: from synergy3DMM import SynergyNet

See DESIGN.md for the scope and kernel design, include/synergy_hip.h for the C ABI.
"""
__all__ = ['SynergyNet', 'ParamsPack']


def __getattr__(name):   # lazy: importing the package must not need a GPU or the built library
    if name == 'SynergyNet':
        from .synergy3DMM import SynergyNet
        return SynergyNet
    if name == 'ParamsPack':
        from .params import ParamsPack
        return ParamsPack
    raise AttributeError(name)
