"""ctypes mirror of include/lumilly_hip.h and include/lumilly_host.h (plain PODs only).

Field order and types must match the headers exactly; tests/test_abi.py checks sizeof() of every
struct against the values the C side reports (lr_host_sizeof / lr_sizeof).
"""
import ctypes as C

LR_ABI_VERSION = 2

LR_OK, LR_EINVAL, LR_EDEVICE, LR_ENOMEM, LR_EUNSUPPORTED, LR_EIO = 0, -1, -2, -3, -4, -5

LR_CAMERA_IDEAL_PINHOLE, LR_CAMERA_THIN_LENS, LR_CAMERA_OMNIDIRECTIONAL = 0, 1, 2
LR_MAT_LAMBERT, LR_MAT_PHONG, LR_MAT_BLINN_PHONG, LR_MAT_GGX, LR_MAT_IDEAL_REFRACTION = 0, 1, 2, 3, 4
LR_PRIM_TRIANGLE, LR_PRIM_SPHERE = 0, 1
LR_SKY_UNIFORM, LR_SKY_IBL = 0, 1
LR_INTEGRATOR_PT, LR_INTEGRATOR_PT_DIRECT = 0, 1
LR_FLAG_PROFILE, LR_FLAG_COUNT, LR_FLAG_STREAMING, LR_FLAG_RESIDENT, LR_FLAG_FUSED = 1, 2, 4, 8, 16
LR_K_GENERATE, LR_K_TRACE, LR_K_SHADE, LR_K_SHADOW, LR_K_RESOLVE, LR_K_RESIDENT, LR_K_PATH, LR_K_COUNT = 0, 1, 2, 3, 4, 5, 6, 7
LR_KERNEL_NAMES = ["generate", "trace", "shade", "shadow", "resolve", "resident", "path"]

f32 = C.c_float
i32 = C.c_int32
u32 = C.c_uint32
u64 = C.c_uint64


class LrCamera(C.Structure):
    _fields_ = [
        ("type", i32),
        ("resolution", i32 * 2),
        ("forward", f32 * 3), ("right", f32 * 3), ("up", f32 * 3),
        ("position", f32 * 3),
        ("aperture_position", f32 * 3),
        ("sensor_size", f32 * 2),
        ("aperture_sensor_distance", f32),
        ("aperture_radius", f32),
        ("focus_distance", f32),
        ("sensor_pixel_area", f32),
        ("sensor_sensitivity", f32),
    ]


class LrMaterial(C.Structure):
    _fields_ = [("type", i32), ("color", f32 * 3), ("emission", f32 * 3), ("param", f32 * 3)]


class LrPrimitive(C.Structure):
    _fields_ = [("type", i32), ("material", i32), ("v", f32 * 9), ("pad", f32)]


class LrSky(C.Structure):
    _fields_ = [
        ("type", i32), ("color", f32 * 3), ("height", i32), ("longitude_offset", f32),
        ("texels", C.POINTER(f32)),
    ]


class LrBvhNode(C.Structure):
    _fields_ = [("x", f32 * 4), ("y", f32 * 4), ("z", f32 * 4), ("child", i32 * 2), ("pad", i32 * 2)]


class LrSceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", u32),
        ("camera", LrCamera),
        ("n_materials", i32),
        ("materials", C.POINTER(LrMaterial)),
        ("n_prims", i32),
        ("prims", C.POINTER(LrPrimitive)),
        ("sky", LrSky),
        ("n_bvh_nodes", i32),
        ("bvh_nodes", C.POINTER(LrBvhNode)),
        ("bvh_prim_order", C.POINTER(i32)),
        ("bvh_max_depth", i32),
    ]


class LrRenderParams(C.Structure):
    _fields_ = [
        ("integrator", i32), ("spp", i32), ("seed", u32), ("depth", i32), ("depth_limit", i32),
        ("no_direct_emitter", i32), ("path_slots", i32), ("flags", i32),
    ]


class LrTile(C.Structure):
    _fields_ = [("x0", i32), ("y0", i32), ("w", i32), ("h", i32)]


class LrStats(C.Structure):
    _fields_ = [
        ("samples", u64), ("segments", u64), ("shadow_rays", u64), ("node_visits", u64),
        ("prim_tests", u64), ("shadow_node_visits", u64), ("shadow_prim_tests", u64),
        ("sky_fetches", u64), ("iterations", u64),
        ("kernel_launches", u64 * LR_K_COUNT),
        ("kernel_ms", C.c_double * LR_K_COUNT),
        ("kernel_timed", u64 * LR_K_COUNT),
        ("render_ms", C.c_double),
        ("upload_ms", C.c_double),
        ("bvh_build_ms", C.c_double),
        ("path_slots", u64),
        ("pipeline", u64),
    ]


# ---- include/lumilly_host.h -------------------------------------------------------------------
class LrRendererConfig(C.Structure):
    _fields_ = [
        ("samples", i32), ("depth", i32), ("depth_limit", i32), ("no_direct_emitter", i32),
        ("threads", i32), ("integrator", i32),
    ]


class LrFilmConfig(C.Structure):
    _fields_ = [("resolution", i32 * 2), ("output", i32), ("gamma", f32), ("sensitivity", f32 * 3)]
