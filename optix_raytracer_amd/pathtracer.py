"""Host side of the path-tracer sample above the C ABI, mirroring the state functions of
SDK/optixPathTracer/optixPathTracer.cpp (same names, same order of calls, same error behaviour:
failures raise, like OPTIX_CHECK/CUDA_CHECK):

    createContext :555 -> buildMeshAccel :576 -> createSBT :829 -> initLaunchParams :424
    -> handleCameraUpdate :448 -> launchSubframe :488

createModule / createProgramGroups / createPipeline (:686-826) have no counterpart: the programs are
fixed-function HIP kernels inside libmipt.so.  torch supplies device memory and streams only.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import api
from .camera import Camera


class OutputBuffer:
    """uchar4 frame buffer with the CUDAOutputBuffer<uchar4> API (SDK/sutil/CUDAOutputBuffer.h:54-94),
    CUDA_DEVICE mode: map() -> device pointer, unmap() -> stream sync, getHostPointer() -> D2H copy."""

    def __init__(self, width, height, device):
        self._device = device
        self._stream = None
        self.resize(width, height)

    def resize(self, width, height):
        self._w, self._h = int(width), int(height)
        self._buf = torch.zeros((self._h, self._w, 4), dtype=torch.uint8, device=self._device)

    def setStream(self, stream): self._stream = stream
    def width(self): return self._w
    def height(self): return self._h
    def map(self): return self._buf.data_ptr()

    def unmap(self):
        (self._stream or torch.cuda.current_stream(self._device)).synchronize()

    def getHostPointer(self):
        return self._buf.cpu().numpy()


class PathTracerState:
    def __init__(self):
        self.context = None
        self.gas_handle = 0
        self.d_vertices = None       # torch tensor (n*3, 4) f32 on device -- caller-owned, read by the shade kernel
        self.stream = None
        self.params = L.MiptParams()
        self.accum = None            # torch tensor (h, w, 4) f32
        self.scene = None
        self.device = None


def createContext(state, device_index=0, log=None, log_level=0):
    state.device = torch.device("cuda", device_index)
    torch.cuda.set_device(state.device)
    state.context = api.Context(device_index, log=log, log_level=log_level)


def buildMeshAccel(state, scene):
    """copy mesh data to device, build the triangle acceleration structure (:576-684)"""
    state.scene = scene
    verts = np.ascontiguousarray(scene["vertices"], np.float32)
    state.d_vertices = torch.from_numpy(verts).to(state.device)
    d_mat = torch.from_numpy(np.ascontiguousarray(scene["mat_index"]).astype(np.int32)).to(state.device)
    state.gas_handle = state.context.accel_build([{
        "vertices": state.d_vertices.data_ptr(), "num_vertices": verts.shape[0], "stride": 16,
        "sbt_index_offsets": d_mat.data_ptr(),
    }])
    del d_mat   # the reference frees d_mat_indices right after the build too (:662)


def createSBT(state):
    sc = state.scene
    state.context.sbt_set_miss(sc["bg"])
    recs = [(sc["emission"][i], sc["diffuse"][i], state.d_vertices.data_ptr()) for i in range(sc["emission"].shape[0])]
    state.context.sbt_set_hitgroups(recs)


def initLaunchParams(state, width, height, samples_per_launch):
    p = state.params
    p.width, p.height = width, height
    state.accum = torch.zeros((height, width, 4), dtype=torch.float32, device=state.device)
    p.accum_buffer = state.accum.data_ptr()
    p.frame_buffer = None
    p.samples_per_launch = samples_per_launch
    p.subframe_index = 0
    lt = state.scene["light"]
    for k in ("corner", "v1", "v2", "normal", "emission"):
        setattr(p.light, k, (C.c_float * 3)(*[float(x) for x in lt[k]]))
    p.handle = state.gas_handle
    state.stream = torch.cuda.Stream(device=state.device)
    # (the accumulation buffer was cleared on torch's current stream; a launch writes pixels as soon as their paths end)
    state.stream.wait_stream(torch.cuda.current_stream(state.device))


def handleCameraUpdate(state, camera):
    p = state.params
    camera.setAspectRatio(np.float32(p.width) / np.float32(p.height))
    p.eye = (C.c_float * 3)(*[float(x) for x in camera.eye()])
    U, V, W = camera.UVWFrame()
    p.U = (C.c_float * 3)(*U.tolist())
    p.V = (C.c_float * 3)(*V.tolist())
    p.W = (C.c_float * 3)(*W.tolist())


def launchSubframe(output_buffer, state):
    result_buffer_data = output_buffer.map()
    state.params.frame_buffer = result_buffer_data
    output_buffer.setStream(state.stream)
    state.context.launch(state.params, state.params.width, state.params.height, 1, stream=state.stream.cuda_stream)
    output_buffer.unmap()
    state.context.sync()


def cleanupState(state):
    if state.context:
        state.context.destroy()
        state.context = None
    state.d_vertices = None
    state.accum = None


def make_state(scene, width, height, spp=1, device_index=0, max_depth=0, log=None, log_level=0):
    """main() of the sample up to the first launchSubframe (:979-1069)."""
    st = PathTracerState()
    createContext(st, device_index, log, log_level)
    buildMeshAccel(st, scene)
    createSBT(st)
    initLaunchParams(st, width, height, spp)
    cam = scene["camera"]
    handleCameraUpdate(st, Camera(cam["eye"], cam["lookat"], cam["up"], cam["fovY"], 1.0))
    st.context.set_option(L.OPT_MAX_DEPTH, max_depth)
    return st


def renderLoop(state, output_buffer, subframes, on_subframe=None, stats_every=0, stats_out=None):
    """The do { } while loop of main() (optixPathTracer.cpp:1021-1045) without the window: updateState ->
    launchSubframe -> (display callback) -> ++subframe_index, with the three timers of sutil::displayStats
    (sutil.cpp:742-775) accumulated and reported as JSON lines instead of an ImGui overlay."""
    import json
    import time
    state_update_time = render_time = display_time = 0.0
    frames = 0
    for k in range(int(subframes)):
        t0 = time.perf_counter()
        # updateState (:471-485): camera / resize changes would reset subframe_index here
        t1 = time.perf_counter()
        state_update_time += t1 - t0
        t0 = t1
        launchSubframe(output_buffer, state)
        t1 = time.perf_counter()
        render_time += t1 - t0
        t0 = t1
        if on_subframe is not None:
            on_subframe(k, output_buffer)
        t1 = time.perf_counter()
        display_time += t1 - t0
        frames += 1
        if stats_every and frames % stats_every == 0:
            line = {"subframe": int(state.params.subframe_index), "state_update_ms": 1e3 * state_update_time / frames,
                    "render_ms": 1e3 * render_time / frames, "display_ms": 1e3 * display_time / frames}
            (stats_out or print)(json.dumps(line))
        state.params.subframe_index += 1
    state.context.sync()
    return {"state_update_s": state_update_time, "render_s": render_time, "display_s": display_time, "subframes": frames}


def renderToFile(scene, outfile, width=768, height=768, samples_per_launch=16, subframes=1, device_index=0, max_depth=0):
    """main() with --file (optixPathTracer.cpp:1051-1079; defaults 768x768, 16 spp :75, 930-931): renders and writes
    the uchar4 frame with sutil::saveImage's rules (optix_raytracer_amd.image).  Returns the state."""
    from . import image
    st = make_state(scene, width, height, samples_per_launch, device_index, max_depth)
    out = OutputBuffer(width, height, st.device)
    renderLoop(st, out, subframes)
    image.saveImage(outfile, out.getHostPointer(), False)
    return st
