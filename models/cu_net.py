"""Drop-in import path of the reference: `from models.cu_net import create_cu_net` (cu-net.py:22)
resolves to the MI355X-native implementation in cu_net_amd."""
from cu_net_amd.module import CUNet, create_cu_net  # noqa: F401
