"""Import-compatible surface of the reference's `gpflow_vgpmp` package for the ELBO hot path.
Every module re-exports the MI355X-native implementation in `vgpmp_amd.host` (host logic) which calls
libvgpmp_hip.so through the C ABI of include/vgpmp.h.  See INTEGRATION.md."""
