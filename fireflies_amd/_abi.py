"""ctypes mirror of include/ffx.h (structures, prototypes, error handling).

This module only *declares* the C ABI; it loads nothing.  `fireflies_amd._lib` binds it to
libffx_hip.so (the product).  The test-only CPU oracle binds the same declarations to its own
shared library (oracle/oracle.py) — both libraries implement the same header.
"""
import ctypes as C

FFX_MAX_LEVELS = 96
FFX_ABI_VERSION = 10
PATTERN_SYNC_BYTES = 35840  # FFX_PATTERN_SYNC_BYTES
FFX_ERR_UNSUPPORTED = -3
REDUCE_SUM = 0
REDUCE_SOFTOR = 1

c_f = C.c_float
c_i = C.c_int
c_p = C.c_void_p


class BvhInfo(C.Structure):
    _fields_ = [
        ("n_tris", C.c_int32),
        ("n_nodes", C.c_int32),
        ("n_levels", C.c_int32),
        ("max_depth", C.c_int32),
        ("off_nodes", C.c_uint64),
        ("off_order", C.c_uint64),
        ("off_refit", C.c_uint64),
        ("off_recs", C.c_uint64),
        ("total_bytes", C.c_uint64),
        ("level_start", C.c_int32 * (FFX_MAX_LEVELS + 1)),
        ("n_wide", C.c_int32),
        ("wide_depth", C.c_int32),
        ("wide_root", C.c_int32),
        ("wide_pad", C.c_int32),
        ("off_wnodes", C.c_uint64),
        ("off_wsrc", C.c_uint64),
        ("off_tq", C.c_uint64),
        ("off_whdr", C.c_uint64),
        ("off_plan", C.c_uint64),
        ("n_treelets", C.c_int32),
        ("plan_ints", C.c_int32),
        ("off_nrec", C.c_uint64),
        ("off_gn", C.c_uint64),
        ("off_bins", C.c_uint64),
        ("bins_stride", C.c_uint64),
    ]


class Smooth(C.Structure):
    """ffx_smooth: interpolated shading normals (host struct with two host tables and device tables)"""
    _fields_ = [
        ("shape_smooth", C.POINTER(C.c_int32)),
        ("shape_vbase", C.POINTER(C.c_int32)),
        ("adj_start", C.c_void_p),
        ("adj", C.c_void_p),
        ("n_vn", C.c_int32),
        ("vnormals", C.c_void_p),
    ]


class AdamArgs(C.Structure):
    """ffx_adam_args: the update ffx_pattern_bwd_blur applies behind the gradient"""
    _fields_ = [
        ("rays", C.c_void_p),
        ("exp_avg", C.c_void_p),
        ("exp_avg_sq", C.c_void_p),
        ("step", C.c_void_p),
        ("grad_out", C.c_void_p),
        ("counter", C.c_void_p),
        ("lr", C.c_double),
        ("beta1", C.c_double),
        ("beta2", C.c_double),
        ("eps", C.c_double),
        ("KF_inv", C.c_float * 16),
        ("lo", C.c_float),
        ("hi", C.c_float),
        ("grad_div", C.c_float),
        ("n_normalize", C.c_int32),
        ("dot_a", C.c_void_p),
        ("dot_b", C.c_void_p),
        ("dot_n", C.c_int64),
        ("dot_partial", C.c_void_p),
        ("dot_b_n", C.c_int64),
        ("guard", C.c_void_p),
    ]


class RandDraw(C.Structure):
    """ffx_rand_draw: one uniform sampler call of a randomisation"""
    _fields_ = [("n", C.c_int32), ("lo", C.c_float * 4), ("hi", C.c_float * 4), ("pad", C.c_int32 * 3)]


class RandEntity(C.Structure):
    """ffx_rand_entity: how one entity's matrices follow from its draws"""
    _fields_ = [("kind", C.c_int32), ("parent", C.c_int32), ("draw_t", C.c_int32), ("draw_r", C.c_int32), ("draw_s", C.c_int32), ("pad", C.c_int32 * 3),
                ("world", C.c_float * 16), ("centroid", C.c_float * 3), ("pad2", C.c_float)]


class StepOp(C.Structure):
    """ffx_step_op: one compiled key write of a scene sample (include/ffx.h ffx_scene_step_h)"""
    _fields_ = [("kind", C.c_int32), ("src", C.c_int32), ("comp", C.c_int32), ("dst", C.c_int32), ("conv", C.c_int32), ("mode", C.c_int32), ("pad", C.c_int32 * 2)]


STEP_POSE_SD, STEP_VALUE_SD, STEP_VALUE_MAT, STEP_MESH = 0, 1, 2, 3


class StepPlan(C.Structure):
    _fields_ = [("ops", C.POINTER(StepOp)), ("n_ops", C.c_int32), ("n_shapes", C.c_int32), ("n_draws", C.c_int32), ("n_ents", C.c_int32),
                ("frame_base", C.POINTER(C.c_int32)), ("frame_stride", C.POINTER(C.c_int32)), ("n_frames", C.POINTER(C.c_int32)),
                ("n_mat_floats", C.c_int32), ("pad", C.c_int32)]


class StepGeom(C.Structure):
    """ffx_step_geom: the blob a scene sample is re-fitted into and the static device tables of the re-fit"""
    _fields_ = [("bvh", C.c_void_p), ("info", C.POINTER(BvhInfo)), ("src_verts", C.c_void_p), ("tris", C.c_void_p), ("tri_shape", C.c_void_p), ("smooth", C.POINTER(Smooth))]


class Camera(C.Structure):
    _fields_ = [
        ("to_world", c_f * 16),
        ("camera_to_sample", c_f * 16),
        ("near_clip", c_f),
        ("far_clip", c_f),
        ("width", C.c_int32),
        ("height", C.c_int32),
    ]


class Projector(C.Structure):
    _fields_ = [
        ("to_world", c_f * 16),
        ("camera_to_sample", c_f * 16),
        ("scale", c_f),
        ("color", c_f * 3),
        ("tex_w", C.c_int32),
        ("tex_h", C.c_int32),
        ("tex_channels", C.c_int32),
        ("enabled", C.c_int32),
    ]


class Spot(C.Structure):
    _fields_ = [
        ("to_world", c_f * 16),
        ("intensity", c_f * 3),
        ("cutoff_deg", c_f),
        ("beam_width_deg", c_f),
        ("enabled", C.c_int32),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("cam", Camera),
        ("proj", Projector),
        ("spot", Spot),
        ("shadows", C.c_int32),
        ("n_shapes", C.c_int32),
        ("mat_stride", C.c_int32),
        ("n_base_tex", C.c_int32),
        ("base_tex_w", C.c_int32 * 4),
        ("base_tex_h", C.c_int32 * 4),
        ("base_tex", C.c_void_p * 4),
        ("slot_uv", C.c_void_p),
        ("n_mat_h", C.c_int32),
        ("mat_h", c_f * 128),
        ("rfilter", C.c_int32),  # RFILTER_BOX / RFILTER_GAUSSIAN: served by ffx_render_{fwd,bwd}_filtered only
        ("rfilter_stddev", c_f),
    ]


# material rows (include/ffx.h FFX_MAT_*)
MAT_STRIDE = 16
MAT_MODEL, MAT_ROUGHNESS, MAT_ANISOTROPIC, MAT_METALLIC, MAT_SPEC_TRANS, MAT_ETA = 3, 4, 5, 6, 7, 8
MAT_SPEC_TINT, MAT_SHEEN, MAT_SHEEN_TINT, MAT_FLATNESS, MAT_CLEARCOAT, MAT_CLEARCOAT_GLOSS = 9, 10, 11, 12, 13, 14
MAT_BASE_TEX = 15
RENDER_FP16, RENDER_SPARSE_ADJOINT, RENDER_APEX_READY, RENDER_CACHE_ZEROED, RENDER_CACHE_KEEP_DROPPED = 1, 2, 4, 8, 16  # flags in the img_fp16 argument of the render calls
MAX_BASE_TEX = 4
RFILTER_BOX, RFILTER_GAUSSIAN = 0, 1
MAX_MAT_H = 128
ADJOINT_DOT_SLOTS = 4096  # FFX_ADJOINT_DOT_SLOTS: partial sums of <gimg, img> in ffx_render_fwd_adjoint


PF = C.POINTER(c_f)

# name -> (restype, argtypes); every symbol include/ffx.h declares
PROTOTYPES = {
    "ffx_last_error": (C.c_char_p, []),
    "ffx_abi_version": (c_i, []),
    "ffx_backend": (C.c_char_p, []),
    "ffx_project_rays_fwd": (c_i, [c_p, c_i, PF, c_p, c_p]),
    "ffx_project_rays_bwd": (c_i, [c_p, c_i, PF, c_p, c_p, c_p]),
    "ffx_transform_points": (c_i, [c_p, c_i, PF, c_i, c_p, c_p]),
    "ffx_l1_value_grad": (c_i, [c_p, c_p, C.c_long, C.c_float, c_p, c_p, c_p]),
    "ffx_l1_value_grad_acc": (c_i, [c_p, c_p, C.c_long, C.c_float, c_p, c_p, c_p, c_p]),
    "ffx_clamp_to_fov": (c_i, [c_p, c_i, PF, PF, C.c_float, C.c_float, c_i, c_p]),
    "ffx_pattern_ws_floats": (C.c_size_t, [c_i, c_i]),
    "ffx_pattern_fwd": (c_i, [c_p, c_i, PF, c_f, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, C.c_long, c_p]),
    "ffx_pattern_bwd": (c_i, [c_p, c_i, PF, c_f, c_i, c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_p]),
    "ffx_pattern_fwd_blur": (c_i, [c_p, c_i, PF, c_f, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, C.c_long, c_i, c_f, c_p, c_p]),
    "ffx_pattern_bwd_blur": (c_i, [c_p, c_i, PF, c_f, c_i, c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_i, c_f, c_p, C.POINTER(AdamArgs), c_p]),
    # rays, n, KF, sigma, size0, size1, tsum, tsor, gtex, reg_weight, ws, grays_data, grays_reg, reg_value, loss_in, loss_in_n, loss_div, ksize, blur_sigma, adam,
    # pts, zero, n_zero, tex, rays_kept, check_kept, sync, epoch, stream
    "ffx_pattern_step": (c_i, [c_p, c_i, PF, c_f, c_i, c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_i, c_f, C.POINTER(AdamArgs),
                               c_p, c_p, C.c_long, c_p, c_p, c_i, c_p, C.c_uint32, c_p]),
    "ffx_scene_refit_top": (c_i, [c_p, C.POINTER(BvhInfo), c_p]),
    "ffx_adam_clamp_step": (c_i, [c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_i, C.c_double, C.c_double, C.c_double, C.c_double, PF, PF, c_f, c_f, c_i, c_p, c_p]),
    "ffx_splat_dense_fwd": (c_i, [c_p, c_i, c_f, c_i, c_i, c_p, c_p]),
    "ffx_splat_dense_bwd": (c_i, [c_p, c_i, c_f, c_i, c_i, c_p, c_p, c_p]),
    "ffx_splat_fwd": (c_i, [c_p, c_i, c_f, c_i, c_i, c_i, c_i, c_p, c_p]),
    "ffx_splat_bwd": (c_i, [c_p, c_i, c_f, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p]),
    "ffx_splat_depth_fwd": (c_i, [c_p, c_p, c_i, c_f, c_i, c_i, c_p, c_p]),
    "ffx_splat_lines_fwd": (c_i, [c_p, c_i, c_f, c_i, c_i, c_p, c_p]),
    "ffx_splat_lines_bwd": (c_i, [c_p, c_i, c_f, c_i, c_i, c_p, c_p, c_p]),
    "ffx_torch_rand_h": (c_i, [C.c_uint64, C.c_uint64, c_i, PF, C.POINTER(C.c_uint64)]),
    "ffx_torch_rand_batch_h": (c_i, [c_i, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32), PF]),
    "ffx_mat4_mul_h": (c_i, [c_p, c_p, c_p]),
    "ffx_scene_randomize_h": (c_i, [c_i, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(RandDraw), c_i, C.POINTER(RandEntity), c_i, c_p, c_p, c_p, c_p]),
    "ffx_blur_fwd": (c_i, [c_p, c_i, c_i, c_i, c_f, c_p, c_p]),
    "ffx_blur_bwd": (c_i, [c_p, c_i, c_i, c_i, c_f, c_p, c_p]),
    "ffx_rgb_to_gray": (c_i, [c_p, c_i, C.c_size_t, c_f, c_f, c_f, c_p, c_p]),
    "ffx_silhouette_fwd": (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p]),
    "ffx_noise_clamp": (c_i, [c_p, c_p, C.c_size_t, c_f, c_f, c_f, c_f, c_p, c_p]),
    "ffx_bvh_blob_bytes": (C.c_size_t, [c_i]),
    "ffx_bvh_build_host": (c_i, [c_p, c_i, c_p, c_i, c_p, C.c_size_t, C.POINTER(BvhInfo)]),
    "ffx_scene_update": (c_i, [c_p, C.POINTER(BvhInfo), c_p, c_p, c_p, c_p, c_p, c_i, C.POINTER(Smooth), c_p]),
    "ffx_scene_update_h": (c_i, [c_p, C.POINTER(BvhInfo), c_p, c_p, c_p, C.POINTER(C.c_int32), PF, c_i, C.POINTER(Smooth), c_p]),
    "ffx_trace_primary": (c_i, [c_p, C.POINTER(BvhInfo), C.POINTER(Camera), c_i, c_i, C.c_uint32, c_p, c_p, c_p, c_p]),
    "ffx_trace_rays": (c_i, [c_p, C.POINTER(BvhInfo), c_p, c_p, c_i, c_f, c_p, c_p, c_p, c_p]),
    "ffx_render_fwd": (c_i, [c_p, C.POINTER(BvhInfo), C.POINTER(SceneDesc), c_p, c_p, c_i, C.c_uint32, c_i, c_p, c_p]),
    "ffx_render_bwd": (c_i, [c_p, C.POINTER(BvhInfo), C.POINTER(SceneDesc), c_p, c_i, C.c_uint32, c_i, c_p, c_p, c_p]),
    "ffx_render_bwd_det_bytes": (C.c_size_t, [C.POINTER(SceneDesc)]),
    "ffx_render_bwd_det": (c_i, [c_p, C.POINTER(BvhInfo), C.POINTER(SceneDesc), c_p, c_i, C.c_uint32, c_i, c_p, c_p, c_p, c_p]),
    "ffx_render_bwd_det_part": (c_i, [c_p, C.POINTER(BvhInfo), C.POINTER(SceneDesc), c_p, c_i, C.c_uint32, c_i, c_p, c_i, c_i, c_p, c_p, c_p]),
    "ffx_det_scale_log2": (c_i, [C.c_uint32, C.c_uint64]),
    "ffx_det_finish": (c_i, [c_p, c_i, C.c_size_t, c_p, c_p]),
    "ffx_render_cache_bytes": (C.c_size_t, [c_i, c_i, c_i]),
    "ffx_render_cache_bytes_sd": (C.c_size_t, [C.POINTER(SceneDesc), c_i]),
    "ffx_render_fwd_cache": (c_i, [c_p, C.POINTER(BvhInfo), C.POINTER(SceneDesc), c_p, c_p, c_i, C.c_uint32, c_i, c_p, c_p, c_p]),
    "ffx_render_bwd_cached": (c_i, [C.POINTER(SceneDesc), c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_p, c_p]),
    "ffx_render_bwd_cached_l1": (c_i, [C.POINTER(SceneDesc), c_p, c_p, c_i, c_p, c_p, c_f, c_p, c_p, c_p]),  # sd, mats, cache, spp, img, target, weight, gtex, slots, stream
    "ffx_render_dot_slots": (C.c_size_t, [c_i, c_i]),
    "ffx_render_filter_bytes": (C.c_size_t, [C.POINTER(SceneDesc)]),
    "ffx_render_fwd_filtered": (c_i, [c_p, C.POINTER(BvhInfo), C.POINTER(SceneDesc), c_p, c_p, c_i, C.c_uint32, c_i, c_p, c_p, c_p]),
    "ffx_render_fwd_adjoint_filtered": (c_i, [c_p, C.POINTER(BvhInfo), C.POINTER(SceneDesc), c_p, c_p, c_i, C.c_uint32, c_i, c_p, c_p, c_p, c_p, c_p]),
    "ffx_render_bwd_filtered": (c_i, [c_p, C.POINTER(BvhInfo), C.POINTER(SceneDesc), c_p, c_i, C.c_uint32, c_i, c_p, c_p, c_p, c_p]),
    "ffx_render_fwd_cache_filtered": (c_i, [c_p, C.POINTER(BvhInfo), C.POINTER(SceneDesc), c_p, c_p, c_i, C.c_uint32, c_i, c_p, c_p, c_p, c_p]),
    "ffx_render_bwd_cached_filtered": (c_i, [C.POINTER(SceneDesc), c_p, c_p, c_i, C.c_uint32, c_p, c_p, c_p]),
    "ffx_render_fwd_adjoint": (c_i, [c_p, C.POINTER(BvhInfo), C.POINTER(SceneDesc), c_p, c_p, c_i, C.c_uint32, c_i, c_p, c_p, c_p, c_p, c_p]),
    "ffx_apex_prepare": (c_i, [c_p, C.POINTER(BvhInfo), C.POINTER(SceneDesc), c_p]),
    "ffx_scene_step_h": (c_i, [C.POINTER(StepPlan), c_p, c_p, c_p, c_p, C.POINTER(SceneDesc), C.POINTER(SceneDesc), c_p, c_p, c_p, C.POINTER(StepGeom), c_i, c_p]),
    "ffx_render_cache_status": (c_i, [c_p, C.POINTER(C.c_uint32), c_p]),
}


class FFXError(RuntimeError):
    pass


def mat16(m):
    """row-major 4x4 (anything with 16 floats when flattened) -> (c_float*16)."""
    flat = [float(v) for row in m for v in (row if hasattr(row, "__len__") else [row])]
    if len(flat) != 16:
        raise ValueError("expected a 4x4 matrix")
    return (c_f * 16)(*flat)


class Api:
    """Pointer-level bindings for one loaded library.  Every method takes raw addresses
    (ints) for [dev] arguments and raises FFXError with ffx_last_error() on failure."""

    def __init__(self, cdll):
        self.lib = cdll
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(cdll, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        ver = cdll.ffx_abi_version()
        if ver != FFX_ABI_VERSION:
            raise FFXError(f"ABI version mismatch: library {ver}, python {FFX_ABI_VERSION}")
        self.backend = cdll.ffx_backend().decode()

    def check(self, rc, what):
        if rc != 0:
            msg = self.lib.ffx_last_error()
            raise FFXError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")

    def call(self, name, *args):
        self.check(getattr(self.lib, name)(*args), name)

    def call_rc(self, name, *args, allow=()):
        """like call, but the return codes in `allow` are handed back instead of raised (an entry point that may decline a shape)"""
        rc = getattr(self.lib, name)(*args)
        if rc != 0 and rc not in allow:
            self.check(rc, name)
        return rc
