"""swarm_amd -- MI355X-native ensemble n-body integrator hot path (drop-in for Swarm-NG's
`integrator=` plugin path).  Product code: HIP kernels behind include/swarm_hip.h plus the host
mirror of the reference's operator surface.  Importing this package requires the built HIP
library; there is no CPU fallback."""
from . import capi, gpulog
from .config import config, default_config, key_not_found
from .ensemble import DeviceBuffer, DeviceEnsemble, HostEnsemble, generate_ensemble
from .integrator import GpuIntegrator, Integrator, LogManager, plugin_not_found, runtime_error, PLUGINS

__all__ = ["capi", "config", "default_config", "key_not_found", "DeviceBuffer", "DeviceEnsemble",
           "HostEnsemble", "generate_ensemble", "GpuIntegrator", "Integrator", "LogManager",
           "plugin_not_found", "runtime_error", "PLUGINS"]
