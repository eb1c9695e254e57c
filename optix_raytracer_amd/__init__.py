"""optix_raytracer_amd -- MI355X-native wavefront path tracer (libmipt.so) and the host-side mirror of
the reference's SDK/optixPathTracer sample.  The compute path is hand-written HIP in csrc/; this
package is the thin host layer over the C ABI in include/mipt.h."""
from . import _lib  # noqa: F401
