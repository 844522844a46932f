"""Drop-in module name of the reference (`from synergy3DMM import SynergyNet`, reference
README.md:84-89); the implementation lives in synergynet_amd/synergy3DMM.py."""
from synergynet_amd.synergy3DMM import SynergyNet, parse_param_62  # noqa: F401
