"""ctypes binding of libmyolo.so (include/myolo.h).  The library is the product: there is NO fallback --
if it cannot be loaded, or a kernel launch fails, the caller gets an exception."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MYOLO_LIB') or os.path.join(_HERE, 'lib', 'libmyolo.so')      # MYOLO_LIB: A/B a differently built library

F32, F16, U8, I64 = 0, 1, 2, 3
ACT_NONE, ACT_SILU, ACT_SIGMOID = 0, 1, 2
MAX_TAPS = 25
STAT_COPIES = 32        # include/myolo.h MYOLO_STAT_COPIES
DT = {torch.float32: F32, torch.float16: F16, torch.uint8: U8, torch.int64: I64}
TORCH_DT = {F32: torch.float32, F16: torch.float16}


class Tensor(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('n', C.c_int32), ('h', C.c_int32), ('w', C.c_int32), ('c', C.c_int32),
                ('sn', C.c_int64), ('sh', C.c_int64), ('sw', C.c_int64), ('dtype', C.c_int32), ('reserved', C.c_int32)]


class BnBwdSeg(C.Structure):
    _fields_ = [('c0', C.c_int32), ('c1', C.c_int32), ('y', Tensor), ('saved', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p),
                ('dsum', C.c_void_p), ('act', C.c_int32), ('reserved', C.c_int32)]


class BnApplyFold(C.Structure):
    _fields_ = [('y', Tensor), ('dy', Tensor), ('saved', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p), ('dsum', C.c_void_p),
                ('dgamma', C.c_void_p), ('dbeta', C.c_void_p), ('act', C.c_int32), ('reserved', C.c_int32)]


class BnSplit(C.Structure):
    _fields_ = [('c_split', C.c_int32), ('count_scale', C.c_int32), ('gamma2', C.c_void_p), ('beta2', C.c_void_p),
                ('running_mean2', C.c_void_p), ('running_var2', C.c_void_p), ('nbt2', C.c_void_p), ('dgamma2', C.c_void_p),
                ('dbeta2', C.c_void_p)]


class BnFwdFuse(C.Structure):
    """myolo_bn_fwd_fuse (include/myolo.h): the BatchNorm + activation half of myolo_conv_bn_act"""
    _fields_ = [('gamma', C.c_void_p), ('beta', C.c_void_p), ('running_mean', C.c_void_p), ('running_var', C.c_void_p), ('nbt', C.c_void_p),
                ('saved', C.c_void_p), ('eps', C.c_float), ('momentum', C.c_float), ('act', C.c_int32), ('reserved', C.c_int32),
                ('res', Tensor), ('out', Tensor), ('split', C.POINTER(BnSplit)), ('barrier', C.c_void_p)]


class SegSyncDesc(C.Structure):
    _fields_ = [('img', C.c_void_p), ('mask', C.c_void_p), ('H0', C.c_int32), ('W0', C.c_int32), ('flip', C.c_int32), ('ow', C.c_int32),
                ('oh', C.c_int32), ('ksh', C.c_int32), ('ksv', C.c_int32), ('x1', C.c_int32), ('y1', C.c_int32), ('wc', C.c_int32),
                ('hc', C.c_int32), ('reserved', C.c_int32), ('hb', C.c_void_p), ('hk', C.c_void_p), ('vb', C.c_void_p), ('vk', C.c_void_p),
                ('xin', C.c_void_p), ('yin', C.c_void_p), ('out_img', C.c_void_p), ('out_lab', C.c_void_p), ('lab_lut', C.c_void_p)]


class MosaicSrc(C.Structure):
    _fields_ = [('img', C.c_void_p), ('h', C.c_int32), ('w', C.c_int32), ('x1a', C.c_int32), ('y1a', C.c_int32), ('x2a', C.c_int32),
                ('y2a', C.c_int32), ('padw', C.c_int32), ('padh', C.c_int32)]


class MosaicDesc(C.Structure):
    _fields_ = [('src', MosaicSrc * 4), ('nsrc', C.c_int32), ('cw', C.c_int32), ('ch', C.c_int32), ('warp', C.c_int32),
                ('M', C.c_double * 6), ('ow', C.c_int32), ('oh', C.c_int32), ('fliplr', C.c_int32), ('flipud', C.c_int32),
                ('fill', C.c_int32), ('reserved', C.c_int32), ('hsv_lut', C.c_void_p), ('out_chw', C.c_void_p), ('out_hwc', C.c_void_p)]


class ConvDesc(C.Structure):
    _fields_ = [('x', Tensor), ('y', Tensor), ('w', C.c_void_p),
                ('cin_pad', C.c_int32), ('cout_pad', C.c_int32), ('wtaps', C.c_int32),
                ('ntaps', C.c_int32), ('stride', C.c_int32), ('up_shift', C.c_int32),
                ('tap_dy', C.c_int32 * MAX_TAPS), ('tap_dx', C.c_int32 * MAX_TAPS), ('tap_w', C.c_int32 * MAX_TAPS),
                ('scale', C.c_void_p), ('shift', C.c_void_p), ('act', C.c_int32), ('accumulate', C.c_int32),
                ('res', Tensor), ('stats', C.c_void_p), ('det_no', C.c_int32), ('nbnb', C.c_int32), ('bnb', C.POINTER(BnBwdSeg))]


class WgradDesc(C.Structure):
    _fields_ = [('x', Tensor), ('dy', Tensor), ('dw', C.c_void_p), ('db', C.c_void_p),
                ('ntaps', C.c_int32), ('stride', C.c_int32), ('up_shift', C.c_int32),
                ('tap_dy', C.c_int32 * MAX_TAPS), ('tap_dx', C.c_int32 * MAX_TAPS),
                ('ksplit', C.c_int32), ('cout', C.c_int32), ('cin', C.c_int32), ('wg_hint', C.c_int32),
                ('ws', C.c_void_p), ('ws_bytes', C.c_int64)]


class DetLossDesc(C.Structure):
    _fields_ = [('nl', C.c_int32), ('na', C.c_int32), ('no', C.c_int32), ('bs', C.c_int32), ('nt', C.c_int32),
                ('dtype', C.c_int32), ('p', C.c_void_p * 5), ('gp', C.c_void_p * 5), ('ny', C.c_int32 * 5),
                ('nx', C.c_int32 * 5), ('anchors', C.c_void_p), ('targets', C.c_void_p), ('balance', C.c_float * 5),
                ('box', C.c_float), ('obj', C.c_float), ('cls', C.c_float), ('cls_pw', C.c_float), ('obj_pw', C.c_float),
                ('anchor_t', C.c_float), ('gr', C.c_float), ('cp', C.c_float), ('cn', C.c_float),
                ('winner', C.c_void_p), ('ciou', C.c_void_p), ('acc', C.c_void_p), ('out', C.c_void_p),
                ('gp32', C.c_void_p), ('gout', C.c_void_p)]


class SgdHyper(C.Structure):
    _fields_ = [('lr', C.c_float * 8), ('momentum', C.c_float * 8), ('weight_decay', C.c_float * 8),
                ('nesterov', C.c_int32), ('reserved', C.c_int32)]


PROG_MAX_ARGS = 24
QUEUE_SEM_BYTES = 256   # include/myolo.h MYOLO_QUEUE_SEM_BYTES
OP_CALL, OP_CALL_SIDE, OP_JOIN, OP_MEMSET = 0, 1, 2, 3


class ProgOp(C.Structure):
    """include/myolo.h myolo_prog_op: one record of a native launch program (csrc/plan_exec.hip)"""
    _fields_ = [('kind', C.c_int32), ('fn', C.c_int32), ('nargs', C.c_int32), ('cond_val', C.c_int32), ('cond', C.c_uint64),
                ('a', C.c_uint64 * PROG_MAX_ARGS)]


class TinyConvDesc(C.Structure):
    """myolo_tiny_conv_desc (include/myolo.h): 1x1 Conv (+BatchNorm) (+activation) on a map of <= TINY_MAX_PIX pixels, one workgroup"""
    _fields_ = [('x', Tensor), ('z', Tensor), ('out', Tensor), ('w', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p),
                ('running_mean', C.c_void_p), ('running_var', C.c_void_p), ('nbt', C.c_void_p), ('saved', C.c_void_p),
                ('eps', C.c_float), ('momentum', C.c_float), ('act', C.c_int32), ('gx_accumulate', C.c_int32),
                ('gout', Tensor), ('dy', Tensor), ('gx', Tensor), ('dgamma', C.c_void_p), ('dbeta', C.c_void_p)]


TINY_MAX_PIX, TINY_MAX_GROUP = 1024, 4      # MYOLO_TINY_MAX_PIX, MYOLO_TINY_MAX_GROUP


class MyoloError(RuntimeError):
    pass


_lib = None
P = C.c_void_p
TP = C.POINTER(Tensor)
_PROTOS = {
    'myolo_version': (C.c_int, []),
    'myolo_arch': (C.c_char_p, []),
    'myolo_set_option': (C.c_int, [C.c_char_p, C.c_int]),
    'myolo_pack_weight': (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, C.c_int, C.c_int, C.c_int, C.c_int, P, P]),
    'myolo_pack_weights_mt': (C.c_int, [P, P, C.c_int, C.c_int, P]),
    'myolo_focus_pack': (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, TP, P]),
    'myolo_conv': (C.c_int, [C.POINTER(ConvDesc), P]),
    'myolo_conv_dgrad_s2': (C.c_int, [C.POINTER(C.POINTER(ConvDesc)), C.c_int, P]),
    'myolo_conv_dgrad_bn': (C.c_int, [C.POINTER(ConvDesc), C.POINTER(BnApplyFold), P]),
    'myolo_conv_pair': (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), P]),
    'myolo_conv_bn_act_ok': (C.c_int, [C.POINTER(ConvDesc)]),
    'myolo_conv_bn_act': (C.c_int, [C.POINTER(ConvDesc), C.POINTER(BnFwdFuse), P]),
    'myolo_conv_wgrad': (C.c_int, [C.POINTER(WgradDesc), P]),
    'myolo_bn_wgrad_stem_ok': (C.c_int, [C.POINTER(WgradDesc), TP]),
    'myolo_bn_wgrad_stem_ws_bytes': (C.c_int64, []),
    'myolo_bn_wgrad_stem': (C.c_int, [C.POINTER(WgradDesc), TP, TP, P, P, P, C.c_int, P, P, P, C.c_int64, P]),
    'myolo_bn_act_fwd': (C.c_int, [TP, P, P, P, P, P, P, P, C.c_float, C.c_float, C.c_int, TP, TP, P]),
    'myolo_bn_act_bwd_reduce': (C.c_int, [TP, TP, P, P, P, C.c_int, P, P]),
    'myolo_bn_act_bwd_apply': (C.c_int, [TP, TP, P, P, P, C.c_int, P, P, P, TP, TP, C.c_int, P]),
    'myolo_bn_act_fwd_split': (C.c_int, [TP, P, P, P, P, P, P, P, C.c_float, C.c_float, C.c_int, TP, TP, C.POINTER(BnSplit), P]),
    'myolo_bn_act_bwd_reduce_split': (C.c_int, [TP, TP, P, P, P, C.c_int, P, C.POINTER(BnSplit), P]),
    'myolo_bn_act_bwd_apply_split': (C.c_int, [TP, TP, P, P, P, C.c_int, P, P, P, TP, TP, C.c_int, C.POINTER(BnSplit), P]),
    'myolo_bn_act_bwd_fused_ok': (C.c_int, [C.c_int, C.c_int64, C.c_int]),
    'myolo_bn_act_bwd_fused': (C.c_int, [TP, TP, P, P, P, C.c_int, P, P, P, TP, TP, C.c_int, C.POINTER(BnSplit), P, P]),
    'myolo_spp_pool_fwd': (C.c_int, [TP, TP, TP, TP, P, P]),
    'myolo_spp_pool_bwd': (C.c_int, [TP, TP, TP, P, TP, C.c_int, P]),
    'myolo_copy_up_fwd': (C.c_int, [TP, TP, C.c_int, P]),
    'myolo_copy_up_bwd': (C.c_int, [TP, TP, C.c_int, C.c_int, P]),
    'myolo_bilinear_fwd': (C.c_int, [TP, TP, P]),
    'myolo_bilinear_bwd': (C.c_int, [TP, TP, C.c_int, P, P]),
    'myolo_adaptive_avgpool_fwd': (C.c_int, [TP, TP, P, P]),
    'myolo_adaptive_avgpool_bwd': (C.c_int, [TP, TP, C.c_int, P]),
    'myolo_adaptive_avgpool_fwd_multi': (C.c_int, [TP, P, C.c_int, P, P]),
    'myolo_tiny_conv_ok': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'myolo_tiny_conv_fwd': (C.c_int, [C.POINTER(TinyConvDesc), C.c_int, P]),
    'myolo_tiny_conv_bwd': (C.c_int, [C.POINTER(TinyConvDesc), C.c_int, P]),
    'myolo_gate_fwd': (C.c_int, [TP, TP, TP, P]),
    'myolo_gate_bwd': (C.c_int, [TP, TP, TP, TP, C.c_int, P, P]),
    'myolo_gate_mul_fwd': (C.c_int, [TP, TP, TP, P]),
    'myolo_gate_mul_bwd': (C.c_int, [TP, TP, TP, TP, C.c_int, P, P]),
    'myolo_add': (C.c_int, [TP, TP, C.c_int, P]),
    'myolo_fill_zero': (C.c_int, [TP, P]),
    'myolo_cast_from_f32': (C.c_int, [P, TP, P]),
    'myolo_dropout_fwd': (C.c_int, [TP, TP, P, C.c_float, P, P]),
    'myolo_dropout_bwd': (C.c_int, [TP, P, TP, C.c_float, C.c_int, P]),
    'myolo_seg_upsample_fwd': (C.c_int, [TP, P, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, P]),
    'myolo_seg_upsample_bwd': (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, TP, C.c_int, P, P]),
    'myolo_seg_sync_transform': (C.c_int, [C.POINTER(SegSyncDesc), P]),
    'myolo_color_jitter': (C.c_int, [P, C.c_int, C.c_int, P, C.c_float, C.c_float, C.c_float, C.c_int, P, P, P, C.c_int, P, P]),
    'myolo_resize_u8': (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, P, P]),
    'myolo_mosaic_warp': (C.c_int, [C.POINTER(MosaicDesc), P]),
    'myolo_adaptive_avgpool_bwd_multi': (C.c_int, [P, C.c_int, TP, C.c_int, P]),
    'myolo_pyramid_upsample_fwd': (C.c_int, [P, C.c_int, TP, P]),
    'myolo_pyramid_upsample_bwd': (C.c_int, [TP, P, C.c_int, P, P, P]),
    'myolo_seg_upce_fwd_grad': (C.c_int, [TP, C.c_int, C.c_int, P, C.c_int, P, P, P, P]),
    'myolo_seg_upce_ohem_pix': (C.c_int, [TP, C.c_int, C.c_int, P, C.c_int, P, P, P]),
    'myolo_seg_upce_ohem_grad': (C.c_int, [TP, C.c_int, C.c_int, P, C.c_int, P, P, C.c_float, P, P]),
    'myolo_seg_lowgrad_apply': (C.c_int, [P, TP, C.c_int, P, P]),
    'myolo_seg_ce_fwd_grad': (C.c_int, [P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, C.c_int, P, P, P]),
    'myolo_seg_ce_scale': (C.c_int, [P, P, P, P]),
    'myolo_seg_argmax': (C.c_int, [TP, P, C.c_int, C.c_int, C.c_int, P]),
    'myolo_seg_metrics': (C.c_int, [P, C.c_int, P, C.c_int64, C.c_int, P, P]),
    'myolo_detect_unpermute': (C.c_int, [P, C.c_int, C.c_int, C.c_int, TP, P]),
    'myolo_detect_decode': (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                      C.POINTER(C.c_float), P, C.c_int64, C.c_int64, P]),
    'myolo_seg_ce_fwd': (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                   P, C.c_int, P, P, P, P]),
    'myolo_ohem_select': (C.c_int, [P, C.c_int64, C.c_float, P, P, P, P, P, P]),
    'myolo_seg_ce_bwd': (C.c_int, [P, P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                   C.c_int64, C.c_int64, C.c_int64, C.c_int64, P, C.c_int, P, P, P, P, C.c_float, P]),
    'myolo_detloss_fwd': (C.c_int, [C.POINTER(DetLossDesc), P]),
    'myolo_detloss_bwd': (C.c_int, [C.POINTER(DetLossDesc), P]),
    'myolo_mt_sgd': (C.c_int, [P, P, C.c_int, C.c_int, C.POINTER(SgdHyper), P, P, P]),
    'myolo_mt_check_finite': (C.c_int, [P, P, C.c_int, C.c_int, C.c_int, P, P]),
    'myolo_mt_ema': (C.c_int, [P, P, C.c_int, C.c_int, C.c_float, P]),
    'myolo_scaler_update': (C.c_int, [P, P, P, C.c_float, C.c_float, C.c_int, P]),
    'myolo_match_predictions': (C.c_int, [P, C.c_int, P, C.c_int, P, C.c_int, P, P, C.c_int64, P]),
    'myolo_frame_pack': (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P, C.c_int, P, P]),
    'myolo_frame_resize_pack': (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P,
                                          C.c_int, P, P]),
    'myolo_seg_blend': (C.c_int, [P, C.c_int, P, C.c_int, C.c_int, P, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, P, P, P]),
    'myolo_prog_fn_id': (C.c_int, [C.c_char_p]),
    'myolo_prog_fn_nargs': (C.c_int, [C.c_int]),
    'myolo_prog_create': (C.c_void_p, [C.POINTER(ProgOp), C.c_int]),
    'myolo_prog_destroy': (None, [C.c_void_p]),
    'myolo_prog_slot': (C.POINTER(C.c_uint64), [C.c_void_p, C.c_int, C.c_int]),
    'myolo_prog_run': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'myolo_prog_last_op': (C.c_int, [C.c_void_p]),
    'myolo_queue_post': (C.c_int, [P, P]),
    'myolo_queue_wait': (C.c_int, [P, C.c_int, P]),
    'myolo_trace_start': (C.c_int, [C.c_int]),
    'myolo_trace_read': (C.c_int64, [C.c_char_p, C.c_int64]),
    'myolo_nms_ws_bytes': (C.c_int64, [C.c_int, C.c_int]),
    'myolo_nms': (C.c_int, [P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int,
                            C.c_int, C.c_int, P, P, P, P, P, P, C.c_uint64, P, P, C.c_int64, P]),
}


def lib():
    """Load libmyolo.so (once).  Raises MyoloError if it is missing -- build it with `python -m multiyolov5_amd.build`."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MyoloError(f'{LIB_PATH} not found: run `python -m multiyolov5_amd.build` (hipcc --offload-arch=gfx950). '
                             'There is no CPU/PyTorch fallback for the multiyolov5_amd hot path.')
        l = C.CDLL(LIB_PATH)   # torch is imported above: libamdhip64.so.7 resolves to the runtime torch already loaded
        for name, (res, args) in _PROTOS.items():
            f = getattr(l, name)
            f.restype, f.argtypes = res, args
        _lib = l
        # MYOLO_SET="name=value,name=value": myolo_set_option pairs applied at load (A/B of the library's run-time options -- tile variants,
        # thresholds -- from the command line of bench.py; every option keeps its default otherwise)
        for kv in filter(None, os.environ.get('MYOLO_SET', '').replace('+', ',').split(',')):
            k, _, v = kv.partition('=')
            if l.myolo_set_option(k.strip().encode(), int(v)) != 0:
                raise MyoloError(f'MYOLO_SET: unknown option {k!r}')
    return _lib


EINVAL = -22


def check(err, what=''):
    if err != 0:
        raise MyoloError(f'libmyolo {what} failed with error {err}' + (' (invalid argument)' if err == -22 else ''))


def launch_trace():
    """{launch site: count} since myolo_trace_start(1) (include/myolo.h: which kernel variants ran; test infrastructure)"""
    l = lib()
    n = l.myolo_trace_read(None, 0)
    buf = C.create_string_buffer(int(n))
    l.myolo_trace_read(buf, n)
    out = {}
    for line in buf.value.decode().splitlines():
        c, _, site = line.partition('\t')
        out[site] = int(c)
    return out


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(t):
    if not t.is_cuda:
        raise MyoloError('multiyolov5_amd runs on MI355X (gfx950) only: tensor is on %s; there is no CPU path' % t.device)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
