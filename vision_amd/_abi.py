"""ctypes mirror of include/vmk.h and include/vmk_host.h (plain C structs, no torch types)."""
import ctypes as C

ABI_VERSION = 7
SLOT_TINTED = 1 << 22  # vmk_slot.tex: image x constant ("multiply" shader node)
SLOT_SPD = 0xFFFFFFFD
SPECTRUM_SRGB, SPECTRUM_HERO = 0, 1
RGB2SPEC_RES = 64
INVALID = 0xFFFFFFFF
MAX_SLOTS = 18
LUT_RES = 32
FILTER_TABLE_SIZE = 20

f32, u32, u64, i32 = C.c_float, C.c_uint32, C.c_uint64, C.c_int32


class Slot(C.Structure):
    _fields_ = [("v", f32 * 3), ("tex", u32)]


class Material(C.Structure):
    _fields_ = [("type", u32), ("flags", u32), ("child0", u32), ("child1", u32), ("slot", Slot * MAX_SLOTS), ("normal", Slot)]


class Light(C.Structure):
    _fields_ = [("type", u32), ("inst_id", u32), ("two_sided", u32), ("scale", f32), ("color", Slot),
                ("alias_offset", u32), ("alias_count", u32), ("alias_integral", f32), ("cond_offset", u32),
                ("res_x", u32), ("res_y", u32), ("w2o", f32 * 9), ("o2w", f32 * 9), ("world_diameter", f32),
                ("position", f32 * 3), ("direction", f32 * 3), ("cos_angle", f32), ("cos_falloff_start", f32),
                ("w2o4", f32 * 16), ("tan_xy", f32 * 2)]


class TriPos(C.Structure):
    _fields_ = [("p0", f32 * 3), ("p1", f32 * 3), ("p2", f32 * 3), ("inst", u32), ("prim", u32), ("pad", u32)]


class TriAttr(C.Structure):
    _fields_ = [("n0", f32 * 3), ("n1", f32 * 3), ("n2", f32 * 3), ("uv0", f32 * 2), ("uv1", f32 * 2),
                ("uv2", f32 * 2), ("pad", f32)]


class Instance(C.Structure):
    _fields_ = [("mat_id", u32), ("light_id", u32), ("tri_offset", u32), ("tri_count", u32), ("n2w", f32 * 9),
                ("o2w", f32 * 16), ("inside_medium", u32), ("outside_medium", u32)]


class Medium(C.Structure):
    _fields_ = [("sigma_a", f32 * 3), ("sigma_s", f32 * 3), ("g", f32), ("scale", f32)]


class Texture(C.Structure):
    _fields_ = [("offset", u64), ("width", u32), ("height", u32), ("format", u32), ("channels", u32)]


class Luts(C.Structure):
    _fields_ = [(n, C.POINTER(f32)) for n in
                ("pure_reflection", "dielectric", "dielectric_inv", "specular", "coat", "sheen_approx", "sheen_volume")]


class Scene(C.Structure):
    _fields_ = [("abi_version", u32), ("n_tris", u32), ("n_instances", u32), ("n_materials", u32), ("n_lights", u32),
                ("n_textures", u32), ("n_alias", u32),
                ("tri_pos", C.POINTER(TriPos)), ("tri_attr", C.POINTER(TriAttr)), ("instances", C.POINTER(Instance)),
                ("materials", C.POINTER(Material)), ("lights", C.POINTER(Light)), ("textures", C.POINTER(Texture)),
                ("tex_data", C.POINTER(C.c_uint8)), ("tex_bytes", u64),
                ("alias_prob", C.POINTER(f32)), ("alias_idx", C.POINTER(u32)), ("alias_func", C.POINTER(f32)),
                ("env_light", u32), ("world_min", f32 * 3), ("world_max", f32 * 3), ("luts", Luts),
                ("n_mediums", u32), ("mediums", C.POINTER(Medium)),
                ("light_alias_offset", u32), ("light_alias_integral", f32),
                ("spectrum", u32), ("rgb2spec", C.POINTER(f32)), ("spd_data", C.POINTER(f32)), ("n_spd", u32),
                ("spd_cie", u32 * 4), ("spd_cie_count", u32), ("spd_cie_interval", f32), ("cie_y_integral", f32), ("spectrum_dimension", u32)]


_T = FILTER_TABLE_SIZE


class RenderParams(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("c2w", f32 * 16), ("raster_to_sensor", f32 * 16),
                ("lens_radius", f32), ("focal_distance", f32), ("filter_type", u32), ("filter_radius", f32 * 2),
                ("filter_marginal_prob", f32 * _T), ("filter_marginal_alias", u32 * _T),
                ("filter_marginal_func", f32 * _T), ("filter_marginal_integral", f32),
                ("filter_cond_prob", f32 * (_T * _T)), ("filter_cond_alias", u32 * (_T * _T)),
                ("filter_cond_func", f32 * (_T * _T)),
                ("max_depth", u32), ("min_depth", u32), ("rr_threshold", f32), ("mis_mode", u32),
                ("env_separate", u32), ("env_prob", f32), ("ray_offset_factor", f32), ("exposure", f32),
                ("tone_mapper", u32), ("process_mediums", u32), ("camera_medium", u32), ("light_sampler", u32)]


COMM_ID_BYTES = 128  # VMK_COMM_ID_BYTES


class Tiles(C.Structure):
    _fields_ = [("tile_size", u32), ("rank", u32), ("world", u32)]


class Counters(C.Structure):
    _fields_ = [(n, u64) for n in ("closest_rays", "shadow_rays", "nodes_visited", "tris_tested", "paths",
                                   "surface_hits", "tex_fetches")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class AccelInfo(C.Structure):
    _fields_ = [("n_nodes", u32), ("n_leaves", u32), ("node_bytes", u32), ("tri_bytes", u32), ("build_ms", f32),
                ("depth", u32), ("stack_depth", u32)]


class HostOptions(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("max_depth", i32), ("min_depth", i32), ("procedural_env", u32),
                ("drop_unsupported_lights", u32), ("lut_path", C.c_char_p), ("mediums", u32), ("spectrum", u32),
                ("missing_assets", u32)]


# every symbol include/vmk.h declares (checked by tests/test_abi.py without a GPU)
VMK_SYMBOLS = ["vmk_create", "vmk_destroy", "vmk_last_error", "vmk_abi_version", "vmk_upload_scene",
               "vmk_build_accel", "vmk_set_render_params", "vmk_set_framebuffer", "vmk_reset_accum",
               "vmk_render_batch", "vmk_synchronize", "vmk_download_accum", "vmk_tonemap", "vmk_get_counters",
               "vmk_reset_counters", "vmk_set_traversal_counters", "vmk_stream", "vmk_accel_info_get", "vmk_trace_rays", "vmk_test_eval",
               "vmk_precompute_albedo", "vmk_render_aov", "vmk_self_check", "vmk_set_auto_self_check", "vmk_tile_skew", "vmk_comm_unique_id", "vmk_comm_init",
               "vmk_comm_adopt", "vmk_enable_kernel_timing", "vmk_collect_kernel_ms", "vmk_allreduce_framebuffer", "vmk_allgather_framebuffer", "vmk_comm_synchronize"]
HOST_SYMBOLS = ["vmk_host_register_image", "vmk_host_clear_images", "vmk_host_list_images", "vmk_host_load_scene",
                "vmk_host_free_scene", "vmk_host_scene_tables", "vmk_host_render_params", "vmk_host_output_spp",
                "vmk_host_output_fn", "vmk_host_describe", "vmk_host_last_error", "vmk_host_build_rgb2spec",
                "vmk_host_load_image", "vmk_host_free_image", "vmk_host_final_picture_mode", "vmk_host_save_image"]
