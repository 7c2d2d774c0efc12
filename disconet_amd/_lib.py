"""ctypes binding of libdisconet_hip.so (include/disconet_hip.h, disconet_train.h, disconet_seg.h).

There is no fallback: if the shared object is missing or a call fails, this
module raises.  The library is built in-tree by disconet_amd/csrc/build.py
(`python -m disconet_amd.csrc.build` or `__graft_entry__.build()`).
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_long, c_size_t,
                    c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
# DISCONET_HIP_LIB: load another build of the same library (A/B runs of a kernel variant)
LIB_PATH = os.environ.get("DISCONET_HIP_LIB") or os.path.join(_HERE, "libdisconet_hip.so")


class DnError(RuntimeError):
    pass


class ConvDesc(Structure):
    """struct dn_conv_desc"""
    _fields_ = [(n, c_int32) for n in (
        "n_images", "h_in", "w_in", "c0", "c1", "up0", "c_out", "ksize", "stride", "relu",
        "ld0", "ld1", "ldo", "math")]


class PackJob(Structure):
    """struct dn_pack_job"""
    _fields_ = [("desc", ConvDesc), ("weight", c_void_p), ("packed", c_void_p), ("mode", c_int32), ("cin_total", c_int32),
                ("ci_first", c_int32), ("n_in", c_int32), ("wmul", c_float), ("reserved", c_int32)]


class FoldJob(Structure):
    """struct dn_fold_job"""
    _fields_ = [("partials", c_void_p), ("sums", c_void_p), ("out", c_void_p), ("n_blocks", c_int32), ("c", c_int32),
                ("accumulate", c_int32), ("reserved", c_int32)]


class Post1x1Desc(Structure):
    """struct dn_post1x1_desc"""
    _fields_ = [(n, c_int32) for n in ("c_out2", "relu2", "split", "ldo_a", "ldo_b", "block_diag")]


class FuseMlpParams(Structure):
    """struct dn_fuse_mlp_params"""
    _fields_ = [(n, c_void_p) for n in ("packed", "s1", "t1", "s2", "t2", "s3", "t3", "w4", "b4")]


class MlpTailParams(Structure):
    """struct dn_mlp_tail_params"""
    _fields_ = [(n, c_void_p) for n in (
        "bn1_scale", "bn1_shift", "w2", "s2", "t2", "w3", "s3", "t3", "w4", "b4")]


# name -> (restype, argtypes); must list every symbol include/*.h declares
SIGNATURES = {
    "dn_version": (c_int, []),
    "dn_build_id": (c_char_p, []),
    "dn_last_error": (c_char_p, []),
    "dn_sp_range_flags": (ctypes.c_uint, [c_int]),
    "dn_sp_range_flags_async": (c_int, [c_void_p, c_int, c_void_p]),
    "dn_voxelize_occupy": (c_int, [c_void_p, c_int, c_int, POINTER(c_double), POINTER(c_double),
                                   POINTER(c_int), c_void_p, c_void_p]),
    "dn_voxel_compact_workspace": (c_size_t, [POINTER(c_int)]),
    "dn_voxel_compact": (c_int, [c_void_p, POINTER(c_int), c_void_p, c_int, c_void_p, c_void_p,
                                 c_void_p]),
    "dn_scatter_dense": (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(c_int), c_void_p,
                                 c_void_p]),
    "dn_scatter_dense_sp": (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(c_int), c_void_p,
                                    c_void_p]),
    "dn_scatter_dense_sp_hi": (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(c_int), c_void_p,
                                       c_void_p]),
    "dn_scatter_dense_bits": (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(c_int), c_void_p,
                                      c_void_p]),
    "dn_conv_packed_weight_floats": (c_size_t, [POINTER(ConvDesc)]),
    "dn_conv_pack_weights": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p]),
    "dn_fold_bn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                           c_void_p, c_void_p, c_void_p]),
    "dn_conv2d": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_void_p, c_void_p]),
    "dn_conv2d_taps": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                               c_long, c_int, c_int, c_void_p]),
    "dn_post1x1_packed_floats": (c_size_t, []),
    "dn_post1x1_pack_weights": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "dn_conv2d_post1x1": (c_int, [POINTER(ConvDesc), POINTER(Post1x1Desc)] + [c_void_p] * 11),
    "dn_sp_tensor_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dn_sp_from_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dn_sp_to_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dn_spconv_packed_weight_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "dn_spconv_pack_weights": (c_int, [POINTER(ConvDesc), c_void_p, c_float, c_void_p, c_void_p]),
    "dn_spconv_pack_multi_table_bytes": (c_size_t, [c_int]),
    "dn_spconv_pack_multi_prepare": (c_int, [POINTER(PackJob), c_int, c_void_p, POINTER(c_int)]),
    "dn_spconv_pack_weights_multi": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "dn_conv_pack_multi_table_bytes": (c_size_t, [c_int]),
    "dn_conv_pack_multi_prepare": (c_int, [POINTER(PackJob), c_int, c_void_p, POINTER(c_int)]),
    "dn_conv_pack_weights_multi": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "dn_spconv2d": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_void_p]),
    "dn_spconv_ks_supported": (c_int, [POINTER(ConvDesc), c_int]),
    "dn_spconv_workspace_bytes": (c_size_t, [POINTER(ConvDesc), c_int]),
    "dn_spconv2d_ks": (c_int, [POINTER(ConvDesc), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "dn_spconv2d_pre_pair_supported": (c_int, [POINTER(ConvDesc), POINTER(ConvDesc)]),
    "dn_spconv2d_pre_pair": (c_int, [POINTER(ConvDesc), POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "dn_spconv2d_dual": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_int, c_void_p]),
    "dn_spconv2d_nhwc": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                 c_void_p]),
    "dn_sp_post1x1_packed_bytes": (c_size_t, []),
    "dn_sp_post1x1_pack_weights": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p]),
    "dn_sp_post1x1_pack_heads": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p, c_void_p]),
    "dn_spconv2d_post1x1": (c_int, [POINTER(ConvDesc), POINTER(Post1x1Desc)] + [c_void_p] * 8 +
                            [c_int, c_void_p, c_void_p, c_void_p]),
    "dn_spconv_force_config": (c_int, [c_int]),
    "dn_spconv_set_upmode": (c_int, [c_int]),
    "dn_decode_boxes": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p, c_void_p,
                                c_void_p]),
    "dn_warp_neighbors": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_int, c_void_p, c_void_p]),
    "dn_warp_neighbors_fm": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_int, c_void_p, c_void_p]),
    "dn_disco_fuse_tail": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   POINTER(MlpTailParams), c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dn_fuse_mlp_supported": (c_int, [c_int]),
    "dn_fuse_mlp_set_waves": (c_int, [c_int]),
    "dn_fuse_mlp_packed_bytes": (c_size_t, [c_int]),
    "dn_fuse_mlp_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_void_p,
                                 c_void_p]),
    "dn_disco_fuse_mlp": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(FuseMlpParams), c_int, c_int,
                                  c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    "dn_disco_fuse_mlp_fm": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(FuseMlpParams), c_int, c_int,
                                     c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    "dn_warp_fm_supported": (c_int, [c_int, c_int, c_int]),
    # ---- include/disconet_seg.h ----
    "dn_sp_maxpool2": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dn_sp_upsample2_bilinear": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dn_seg_label_count": (c_int, [c_void_p, c_long, c_int, c_void_p, c_void_p]),
    "dn_seg_ce_loss": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                               c_void_p]),
    "dn_maxpool2_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dn_maxpool2_nhwc_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dn_upsample2_bilinear_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dn_upsample2_bilinear_nhwc_backward": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    # ---- include/disconet_train.h ----
    "dn_conv_wgrad_workspace": (c_size_t, [POINTER(ConvDesc)]),
    "dn_reduce_workspace_bytes": (c_size_t, [c_int, c_long, c_int]),
    "dn_conv_wgrad": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_int, c_int, c_void_p]),
    "dn_conv_wgrad_sp_supported": (c_int, [POINTER(ConvDesc)]),
    "dn_conv_wgrad_sp_workspace": (c_size_t, [POINTER(ConvDesc)]),
    "dn_conv_wgrad_sp": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                                 c_float, c_void_p]),
    "dn_conv_wgrad_sp_z": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                                 c_float, c_void_p]),
    "dn_conv_dgrad_weights": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                      c_void_p]),
    "dn_conv_dgrad_class_weights": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                            POINTER(c_int), c_void_p]),
    "dn_bn_train_stats": (c_int, [c_void_p, c_int, c_long, c_int, c_int, c_void_p, c_size_t, c_void_p,
                                  c_void_p, c_void_p]),
    "dn_bn_train_stats_partial": (c_int, [c_void_p, c_int, c_long, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "dn_bn_train_stats_finish": (c_int, [c_void_p, c_int, c_long, c_int, c_void_p, c_void_p, c_void_p]),
    "dn_bn_train_backward_partial": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p,
                                             c_void_p, c_int, c_void_p]),
    "dn_bn_train_backward_finish": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_long,
                                            c_void_p, c_void_p]),
    "dn_bn_train_backward_finish_sp": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_long,
                                               c_void_p, c_void_p, c_float, c_void_p]),
    "dn_bn_train_stats_running": (c_int, [c_void_p, c_long, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_float, c_void_p]),
    "dn_bn_bias_workspace_bytes": (c_size_t, [c_long, c_int]),
    "dn_bn_train_backward_finish_bias": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_void_p, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p, c_long,
                                                 c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dn_bn_train_backward_finish_bias_deferred": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                                          c_void_p, c_void_p, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                                          c_long, c_void_p, c_void_p, c_float, c_void_p, c_size_t, POINTER(c_int),
                                                          c_void_p]),
    "dn_channel_sum_partial": (c_int, [c_void_p, c_long, c_int, c_int, c_void_p, c_size_t, POINTER(c_int), c_void_p]),
    "dn_channel_sum_fold_multi": (c_int, [POINTER(FoldJob), c_int, c_void_p]),
    "dn_bn_train_apply": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                  c_int, c_long, c_int, c_int, c_void_p, c_void_p]),
    "dn_bn_train_apply_mask": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_long, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p]),
    "dn_bn_train_apply_mask_sp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_long, c_int, c_int, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p]),
    "dn_bn_update_running": (c_int, [c_void_p, c_void_p, c_int, c_long, c_int, c_void_p, c_float,
                                     c_void_p, c_void_p, c_void_p]),
    "dn_bn_train_backward": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int,
                                     c_int, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                                     c_int, c_void_p]),
    "dn_channel_sum": (c_int, [c_void_p, c_long, c_int, c_int, c_void_p, c_size_t, c_void_p, c_int, c_void_p]),
    "dn_upsample2_sum": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dn_add_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_long, c_int, c_void_p]),
    "dn_pair_add_ego": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dn_pair_sum_ego": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                c_void_p]),
    "dn_fuse_combine": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "dn_fuse_combine_backward": (c_int, [c_void_p, c_int] + [c_void_p] * 7 + [c_int, c_int, c_int,
                                                                             c_void_p, c_void_p,
                                                                             c_void_p]),
    "dn_warp_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p]),
    "dn_warp_list": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                             c_void_p]),
    "dn_det_loss": (c_int, [c_void_p] * 5 + [c_long, c_int, c_float, c_float, c_float, c_float,
                                             c_void_p, c_void_p, c_void_p, c_void_p]),
    "dn_kd_kl_loss": (c_int, [c_void_p, c_void_p, c_long, c_int, c_float, c_void_p, c_void_p, c_int,
                              c_void_p]),
    "dn_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float,
                             c_float, c_float, c_float, c_int, c_void_p]),
}

_lib = None


def load():
    """Load the shared object (once) and bind every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    # A library that was not built from THIS tree must not run (round 4 shipped one): the id baked into it is the hash of
    # every source, header and flag (csrc/build.py), read from the FILE before anything is mapped.  Missing or stale and the
    # sources + hipcc are here: rebuilt, loudly, under a file lock (several processes may import at once) -- what runs is always
    # what the tree says.  DISCONET_NO_AUTOBUILD=1: raise instead.  A/B variants (tools/ab) carry other flags and therefore
    # other ids: DISCONET_HIP_LIB / DISCONET_ALLOW_STALE_LIB=1 say so explicitly and skip the check.
    variant = bool(os.environ.get("DISCONET_HIP_LIB")) or os.environ.get("DISCONET_ALLOW_STALE_LIB") == "1"
    want = None
    if not variant:
        from .csrc import build as _build
        have = _build.built_id(LIB_PATH)
        if not _build.sources_present():
            # a deployment without csrc/*.hip (the built library shipped as an artefact): nothing to compare with or rebuild
            # from -- the library's own baked id stands; a library without one (or none at all) is still refused
            if have is None:
                raise DnError("libdisconet_hip.so is missing or carries no build id (%s), and this tree has no sources to "
                              "build it from. There is no CPU fallback." % LIB_PATH)
            want = have
        else:
            want = _build.tree_hash()
        if have != want:
            why = ("libdisconet_hip.so is not built (%s)" % LIB_PATH if have is None and not os.path.exists(LIB_PATH) else
                   "libdisconet_hip.so is stale: built from tree %s, the sources here hash to %s" % (have, want))
            if os.environ.get("DISCONET_NO_AUTOBUILD") == "1" or os.path.abspath(LIB_PATH) != os.path.abspath(_build.LIB_PATH):
                raise DnError(why + ". Run `python -m disconet_amd.csrc.build` (needs hipcc); there is no CPU fallback.")
            import sys
            print("disconet_amd: %s -- rebuilding with hipcc (python -m disconet_amd.csrc.build)" % why, file=sys.stderr, flush=True)
            try:
                _build.build_locked(verbose=False)
            except Exception as e:      # noqa: BLE001 -- no hipcc, a compile error: never run the stale binary
                raise DnError("%s, and the rebuild failed (%r). There is no CPU fallback." % (why, e)) from e
    elif not os.path.exists(LIB_PATH):
        raise DnError("libdisconet_hip.so variant %s does not exist" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing: loud by design
        fn.restype = restype
        fn.argtypes = argtypes
    if not variant and lib.dn_build_id().decode() != want:
        raise DnError("libdisconet_hip.so reports build id %s, the tree hashes to %s" % (lib.dn_build_id().decode(), want))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().dn_last_error()
        raise DnError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
