"""ctypes loader for libj2kgfx.so -- the HIP implementation behind include/j2kgfx.h.

No fallback of any kind: a missing library or a missing GPU raises.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("J2K_LIB") or os.path.join(os.path.dirname(_PKG), "libj2kgfx.so")   # J2K_LIB: dev override (A/B builds)

OK = 0
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_GO_PANIC, ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6
BAND_LL, BAND_HL, BAND_LH, BAND_HH = 0, 1, 2, 3
CODER_MQ, CODER_HT = 0, 1


class J2KError(RuntimeError):
    def __init__(self, status, msg=""):
        self.status = status
        super().__init__("j2kgfx status %d: %s" % (status, msg))


class Block(C.Structure):
    _fields_ = [("plane", C.c_int32), ("band", C.c_int32), ("x0", C.c_int32), ("y0", C.c_int32),
                ("w", C.c_int32), ("h", C.c_int32)]


class Params(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "ncomp", "precision", "is_signed", "lossless", "quality", "num_resolutions",
        "cb_w", "cb_h", "tile_w", "tile_h", "coder", "tile_first", "tile_count", "frame_rows", "closed_loop")]


class PlanInfo(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "tiles", "planes", "blocks", "coeff_elems", "bytes_cap", "dwt_bytes", "dwt_level0_bytes",
        "block_samples", "decoded_elems")]


# every symbol include/j2kgfx.h declares (tests/test_abi_symbols.py checks the header against this list)
SYMBOLS = [
    "j2k_ctx_create", "j2k_ctx_destroy", "j2k_ctx_sync", "j2k_ctx_stream", "j2k_ctx_last_error",
    "j2k_status_string", "j2k_version", "j2k_ctx_set_option", "j2k_ctx_profile_enable", "j2k_ctx_profile_read", "j2k_ctx_profile_read_tag",
    "j2k_ctx_capture_begin", "j2k_ctx_capture_end", "j2k_graph_launch", "j2k_graph_destroy",
    "j2k_dc_level_shift_forward", "j2k_dc_level_shift_inverse", "j2k_forward_rct", "j2k_inverse_rct",
    "j2k_forward_ict", "j2k_inverse_ict",
    "j2k_forward53", "j2k_inverse53", "j2k_forward97", "j2k_inverse97",
    "j2k_forward2d53", "j2k_inverse2d53", "j2k_forward2d97", "j2k_inverse2d97",
    "j2k_decompose_multilevel53", "j2k_reconstruct_multilevel53",
    "j2k_decompose_multilevel97", "j2k_reconstruct_multilevel97",
    "j2k_tcd_apply_forward_dwt", "j2k_tcd_apply_inverse_dwt",
    "j2k_encode_blocks", "j2k_decode_blocks", "j2k_block_bound",
    "j2k_plan_set_decode_coded_rows_only", "j2k_plan_create", "j2k_plan_destroy", "j2k_plan_get_info", "j2k_plan_get_blocks", "j2k_plan_get_planes",
    "j2k_plan_forward", "j2k_plan_inverse", "j2k_plan_encode_blocks", "j2k_plan_compact", "j2k_plan_encode_stream",
    "j2k_plan_decode_blocks", "j2k_plan_get_decoded_offsets", "j2k_encode_frame",
    "j2k_mq_encode", "j2k_mq_decode", "j2k_raw_encode", "j2k_raw_decode",
    "j2k_convert_colorspace", "j2k_convert_colorspace_device",
    "j2k_pixels_components", "j2k_pixels_precision", "j2k_extract_image_data", "j2k_create_image",
    "j2k_unpack_pixels", "j2k_pack_pixels", "j2k_plan_forward_rgba8", "j2k_plan_inverse_rgba8",
    "j2k_plan_forward_pixels", "j2k_plan_inverse_pixels", "j2k_plan_pixels_fused",
    "j2k_tile_part_bound", "j2k_create_tile_header", "j2k_assemble_tiles", "j2k_read_tile_part_header", "j2k_parse_tile_parts",
    "j2k_plan_tile_parts_bound", "j2k_plan_assemble_tiles_device",
    "j2k_t2_packet_sequence", "j2k_t2_packet_bound", "j2k_t2_encode_packet", "j2k_t2_decode_packet", "j2k_t2_encode_packets_device", "j2k_t2_decode_packets_device", "j2k_plan_t2_packets", "j2k_plan_t2_fill_cbs", "j2k_tagtree_shape", "j2k_tcd_init_tile",
    "j2k_plan_frame_bound", "j2k_plan_encode_tile_parts", "j2k_plan_decode_tile_parts", "j2k_plan_place_blocks", "j2k_plan_frame_status", "j2k_plan_frame_parallel_tiles",
    "j2k_plan_encode_frame_pixels", "j2k_plan_decode_frame_pixels", "j2k_encode_pixels_host", "j2k_decode_pixels_host",
    "j2k_plan_pack_bound", "j2k_plan_pack_stream", "j2k_plan_unpack_stream", "j2k_plan_unpack_streams",
    "j2k_comm_load_error", "j2k_comm_get_unique_id", "j2k_comm_create", "j2k_comm_destroy", "j2k_comm_last_error", "j2k_comm_stream", "j2k_gather_streams", "j2k_comm_wait",
]
T2_FRESH, T2_WIDE_LEN, T2_SEATED = 1, 2, 4          # j2k_t2_dev_packet.flags (closed-loop mode)
PIX_GRAY8, PIX_GRAY16, PIX_RGBA8, PIX_RGBA64, PIX_NRGBA8, PIX_NRGBA64 = range(6)

_lib = None


def lib():
    """Load the shared library (does not touch the GPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise J2KError(ERR_NO_DEVICE, "libj2kgfx.so not built (%s); run `make -C go-jpeg2000_amd` or "
                                          "__graft_entry__.build()" % LIB_PATH)
        # One HIP runtime per process: torch bundles its own libamdhip64 (same soname as
        # /opt/rocm's).  Importing torch FIRST makes libj2kgfx bind to that copy; loading ours
        # first would bring in a second runtime and torch would then see "No HIP GPUs".
        # J2K_LIB_HOST_ONLY=1 (tests/test_sanitized_host_build.py): J2K_LIB is the sanitised build of the host-only sources
        # (t2.cpp, assemble.cpp) -- no HIP in it, so no torch first, and only the symbols it has are given their signatures.
        partial = os.environ.get("J2K_LIB_HOST_ONLY") == "1"
        if not partial:
            try:
                import torch  # noqa: F401
            except ImportError:  # plain C-ABI use without torch is fine (single runtime from /opt/rocm)
                pass
        L = C.CDLL(LIB_PATH)
        V, I, S, I64P, DP = C.c_void_p, C.c_int, C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_double)
        sigs = {
            "j2k_status_string": (C.c_char_p, None), "j2k_version": (C.c_char_p, None), "j2k_ctx_last_error": (C.c_char_p, [V]),
            "j2k_ctx_stream": (V, [V]), "j2k_block_bound": (S, [I, I, I]), "j2k_ctx_create": (I, [I, C.POINTER(V)]),
            "j2k_ctx_destroy": (None, [V]), "j2k_ctx_sync": (I, [V]), "j2k_ctx_profile_enable": (I, [V, I]),
            "j2k_ctx_profile_read": (I, [V, I64P, DP]), "j2k_ctx_profile_read_tag": (I, [V, I, I64P, DP]),
            "j2k_ctx_capture_begin": (I, [V]), "j2k_ctx_capture_end": (I, [V, C.POINTER(V)]), "j2k_graph_launch": (I, [V]),
            "j2k_graph_destroy": (None, [V]), "j2k_plan_destroy": (None, [V]), "j2k_comm_get_unique_id": (I, [V]),
            "j2k_comm_load_error": (C.c_char_p, []), "j2k_comm_create": (I, [V, V, I, I, C.POINTER(V)]), "j2k_comm_destroy": (None, [V]),
            "j2k_comm_last_error": (C.c_char_p, [V]), "j2k_comm_stream": (V, [V]),
            "j2k_gather_streams": (I, [V, I, V, V, V, V, I, V, S, V, I]), "j2k_comm_wait": (I, [V, V]),
            "j2k_plan_pack_bound": (S, [V]), "j2k_ctx_set_option": (I, [V, C.c_char_p, C.c_long]),
        }
        for name, (res, args) in sigs.items():
            if partial and not hasattr(L, name):
                continue
            fn = getattr(L, name)
            fn.restype = res
            if args is not None:
                fn.argtypes = args
        _lib = L
    return _lib
