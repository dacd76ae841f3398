"""musev_amd.models -- MI355X-native stand-ins for musev.models (same module names, constructor flags, forward
signatures and state_dict keys; the arithmetic runs in libmusev_hip.so)."""
from ..utils.register import Register

Model_Register = Register(registry_name="torch_model")
