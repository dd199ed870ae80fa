"""ctypes binding of libuds_hip.so (the C ABI in include/uds_hip.h) for PyTorch-ROCm tensors.

PyTorch is plumbing here: it owns device memory and the stream; every computation below is a
hand-written HIP kernel reached through the C ABI.  There is NO fallback: if the library is
missing or a tensor is not a contiguous fp32 CUDA(HIP) tensor the call raises.
"""
import ctypes
import os

import numpy as np
import torch  # noqa: F401  (must be imported first: its libamdhip64.so.7 is the runtime we bind to)

_HERE = os.path.dirname(os.path.abspath(__file__))
# UDS_LIB_PATH: a differently built copy of the same library (kernel experiments: tools/variant_bench.py)
LIB_PATH = os.environ.get('UDS_LIB_PATH') or os.path.join(_HERE, 'libuds_hip.so')

ABI_VERSION = 22
FLAG_EXACT_FP32, FLAG_REQUIRE_FUSED = 1, 2
PRECISION_FLAGS = {'bf16x3': 0, 'fp32': FLAG_EXACT_FP32}

ACT = {None: 0, 'linear': 0, 'relu': 1, 'tanh': 2, 'sigmoid': 3, 'hard_sigmoid': 4}

_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int
_c_ptr = ctypes.c_void_p

# name -> (restype, argtypes): every symbol include/uds_hip.h declares
SYMBOLS = {
    'uds_abi_version': (_c_int, []),
    'uds_last_error': (ctypes.c_char_p, []),
    'uds_csr_create': (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, ctypes.POINTER(_c_ptr)]),
    'uds_csr_destroy': (_c_int, [_c_ptr]),
    'uds_csr_shape': (_c_int, [_c_ptr, ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i64),
                               ctypes.POINTER(ctypes.c_int32)]),
    'uds_csr_row_order': (_c_int, [_c_ptr, _c_ptr]),
    'uds_dense_act': (_c_int, [_c_ptr, _c_i64, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_i64, _c_int, _c_ptr, _c_ptr,
                               _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    'uds_csr_spmm': (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_int, _c_ptr, _c_ptr]),
    'uds_conv1d_causal': (_c_int, [_c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr,
                                   _c_ptr]),
    'uds_dense_cumsum_heads': (_c_int, [_c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr]),
    'uds_recurrent_fused': (_c_int, [_c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr]),
    'uds_recurrent_fused_supported': (_c_int, [_c_i64, _c_int]),
    'uds_recurrent_forward': (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr]),
    'uds_attn_sum_pool': (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr]),
    'uds_dropout': (_c_int, [_c_ptr, _c_i64, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, _c_ptr, _c_ptr]),
    'uds_recurrent_forward_train': (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr]),
    'uds_recurrent_backward': (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr]),
    'uds_rowgemm_packed_bytes': (_c_i64, [_c_i64, _c_i64]),
    'uds_rowgemm_pack': (_c_int, [_c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr]),
    'uds_gat_aggregate_masked': (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr]),
    'uds_gat_aggregate_coef': (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr]),
    'uds_diffusion_forward': (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr]),
    'uds_halo_pack': (_c_int, [_c_ptr, _c_i64, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_ptr]),
    'uds_halo_unpack': (_c_int, [_c_ptr, _c_i64, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr]),
    'uds_remainder_packed_bytes': (_c_i64, [_c_i64, _c_i64]),
    'uds_remainder_pack': (_c_int, [_c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr]),
    'uds_remainder_workspace_bytes': (_c_i64, [_c_i64, _c_i64, _c_i64, _c_i64]),
    'uds_remainder_forward_dense': (_c_int, [_c_ptr, _c_i64, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_int, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr]),
    'uds_remainder_forward': (_c_int, [_c_ptr, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr]),
    'uds_rowgemm_forward': (_c_int, [_c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr]),
    'uds_rowgemm_forward_pair': (_c_int, [_c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64,
                                          _c_i64, _c_i64, _c_int, _c_ptr]),
    'uds_rowgemm_forward_cat': (_c_int, [_c_ptr, _c_i64, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int,
                                         _c_ptr, _c_i64, _c_i64, _c_ptr]),
    'uds_dense_cumsum': (_c_int, [_c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr]),
    'uds_cumsum_act': (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr]),
    'uds_flow_balance': (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    'uds_gat_workspace_floats': (_c_i64, [_c_i64, _c_i64, _c_i64]),
    'uds_gat_forward': (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64,
                                 _c_int, _c_ptr, _c_ptr, _c_ptr]),
    'uds_gat_aggregate': (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr]),
    'uds_gat_backward': (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr,
                                  _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    'uds_gat_backward_coef': (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr,
                                       _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    'uds_csr_sddmm': (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr]),
    'uds_wgrad_workspace_floats': (_c_i64, [_c_i64, _c_i64, _c_i64, _c_int]),
    'uds_wgrad': (_c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    'uds_network_create': (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, ctypes.POINTER(_c_ptr)]),
    'uds_network_destroy': (_c_int, [_c_ptr]),
    'uds_network_prepare': (_c_int, [_c_ptr, _c_i64, _c_i64]),
    'uds_network_plan_info': (_c_int, [_c_ptr, _c_ptr]),
    'uds_tile_plan_create': (_c_int, [_c_ptr] * 8 + [_c_i64, _c_i64] + [ctypes.c_int32] * 4 + [ctypes.POINTER(_c_ptr)]),
    'uds_tile_plan_destroy': (_c_int, [_c_ptr]),
    'uds_tile_plan_sizes': (_c_int, [_c_ptr, ctypes.POINTER(_c_i64), ctypes.POINTER(_c_i64), _c_ptr]),
    'uds_tile_plan_copy': (_c_int, [_c_ptr, _c_ptr, _c_ptr]),
    'uds_tile_plan_blocks': (_c_int, [_c_ptr, _c_ptr, ctypes.POINTER(_c_i64)]),
    'uds_roll_update': (_c_int, [_c_ptr] * 7 + [_c_i64, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    'uds_spatial_workspace_floats': (_c_i64, [_c_ptr, _c_i64, _c_i64, _c_i64]),
    'uds_spatial_packed_bytes': (_c_i64, []),
    'uds_spatial_pack_weights': (_c_int, [_c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr]),
    'uds_spatial_layer_forward': (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_int,
                                           _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    'uds_spatial_layer_forward_rem': (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_int,
                                      _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    'uds_spatial_layer_forward_split': (_c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_i64,
                                                 _c_i64, _c_i64, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
}


class SpatialParams(ctypes.Structure):
    """uds_spatial_params_t"""
    _fields_ = [(n, _c_ptr) for n in ('xe_k', 'xe_b', 'ex_k', 'ex_b', 'ne_n_val', 'ne_e_val',
                                      'gx_k', 'gx_as', 'gx_an', 'gx_b', 'ge_k', 'ge_as', 'ge_an', 'ge_b', 'packed')]


class UdsError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen libuds_hip.so (built by `python -m gnn_uds_amd.build` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('%s is missing: the HIP engine is not built (run `python -m gnn_uds_amd.build`); '
                          'gnn_uds_amd has no CPU fallback' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.uds_abi_version() != ABI_VERSION:
        raise ImportError('libuds_hip.so ABI %d, binding expects %d' % (lib.uds_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = load().uds_last_error()
        raise UdsError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else '?'))


def _dev(t, name, allow_none=False):
    """Device pointer of a contiguous fp32 HIP tensor (the C ABI takes plain pointers)."""
    if t is None:
        if allow_none:
            return None
        raise UdsError('%s is required' % name)
    if not isinstance(t, torch.Tensor):
        raise UdsError('%s must be a torch.Tensor, got %r' % (name, type(t)))
    if not t.is_cuda:
        raise UdsError('%s is on %s: gnn_uds_amd runs on the MI355X only and has no CPU fallback' % (name, t.device))
    if t.dtype != torch.float32:
        raise UdsError('%s must be float32, got %s' % (name, t.dtype))
    if not t.is_contiguous():
        raise UdsError('%s must be contiguous' % name)
    _same_device(t.device.index, name)
    return t.data_ptr()


def _current_device():
    if not torch.cuda.is_available():
        raise UdsError('no HIP device is visible: gnn_uds_amd runs on the MI355X only and has no CPU fallback')
    return torch.cuda.current_device()


def _same_device(index, what):
    """Kernels are enqueued on the CURRENT device's stream and handles allocate on the current device: an operand or a
    handle that lives on another GPU would be reached as peer traffic at best, fault at worst -- refuse it."""
    cur = _current_device()
    if index is not None and index != cur:
        raise UdsError('%s lives on cuda:%d but the current device is cuda:%d: call torch.cuda.set_device(%d) (one process per GPU) '
                       'or wrap the call in `with torch.cuda.device(%d):`' % (what, index, cur, index, index))


def _dev_i32(t, name):
    """Device pointer of a contiguous int32 HIP tensor (index lists of the backward kernels)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda or t.dtype != torch.int32 or not t.is_contiguous():
        raise UdsError('%s must be a contiguous int32 HIP tensor' % name)
    _same_device(t.device.index, name)
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


class CsrHandle:
    """uds_csr_t: a CSR pattern resident on the device, plus its degree-sorted row schedule."""

    def __init__(self, csr):
        lib = load()
        self.n_rows, self.n_cols, self.nnz = int(csr.n_rows), int(csr.n_cols), int(csr.nnz)
        rowptr = np.ascontiguousarray(csr.rowptr, dtype=np.int32)
        col = np.ascontiguousarray(csr.col, dtype=np.int32)
        h = _c_ptr()
        self.device = _current_device()                  # uds_csr_create allocates on the current device
        _check(lib.uds_csr_create(rowptr.ctypes.data, col.ctypes.data, self.n_rows, self.n_cols, self.nnz,
                                  ctypes.byref(h)), 'uds_csr_create')
        self._h = h
        self._rowptr, self._col = rowptr, col
        self._t = None

    def transposed(self, device):
        """(handle of the transposed pattern, perm_t): perm_t[p] (int32 device tensor) = position in THIS pattern's
        row-major order of entry p of the transposed pattern.  Integer bookkeeping for the backward kernels, built once."""
        if self._t is None:
            rows = np.repeat(np.arange(self.n_rows, dtype=np.int64), np.diff(self._rowptr.astype(np.int64)))
            cols = self._col.astype(np.int64)
            order = np.lexsort((rows, cols))                  # by column, then row: the transpose in row-major order
            t_rowptr = np.zeros(self.n_cols + 1, dtype=np.int64)
            np.add.at(t_rowptr, cols + 1, 1)
            t_rowptr = np.cumsum(t_rowptr)

            class _T:      # what CsrHandle.__init__ reads
                pass
            t = _T()
            t.n_rows, t.n_cols, t.nnz = self.n_cols, self.n_rows, self.nnz
            t.rowptr, t.col = t_rowptr.astype(np.int32), rows[order].astype(np.int32)
            self._t = (CsrHandle(t), order.astype(np.int32))
        h, perm = self._t
        if not isinstance(perm, torch.Tensor) or perm.device != torch.device(device):
            perm = torch.as_tensor(np.asarray(perm.cpu() if isinstance(perm, torch.Tensor) else perm), dtype=torch.int32,
                                   device=device)
            self._t = (h, perm)
        return h, perm

    @property
    def ptr(self):
        _same_device(self.device, 'this CSR handle')
        return self._h

    def row_order(self):
        out = np.empty(self.n_rows, dtype=np.int32)
        _check(load().uds_csr_row_order(self._h, out.ctypes.data), 'uds_csr_row_order')
        return out

    def max_degree(self):
        md = ctypes.c_int32()
        _check(load().uds_csr_shape(self._h, None, None, None, ctypes.byref(md)), 'uds_csr_shape')
        return md.value

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h is not None and _lib is not None:
            _lib.uds_csr_destroy(h)


class NetworkHandle:
    """uds_network_t over a gnn_uds_amd.graph.DrainageGraph."""

    def __init__(self, graph):
        lib = load()
        self.graph = graph
        self.adj = CsrHandle(graph.adj)
        self.edge_adj = CsrHandle(graph.edge_adj)
        self.inc_n = CsrHandle(graph.inc_n)
        self.inc_e = CsrHandle(graph.inc_e)
        h = _c_ptr()
        self.device = _current_device()                  # tile plans are uploaded to the current device
        _check(lib.uds_network_create(self.adj.ptr, self.edge_adj.ptr, self.inc_n.ptr, self.inc_e.ptr, ctypes.byref(h)),
               'uds_network_create')
        self._h = h
        self._prepared = {(64, 64)}

    @property
    def ptr(self):
        _same_device(self.device, 'this network handle')
        return self._h

    def prepare(self, fx, fe):
        """Build the tile plans a layer with input widths (fx, fe) needs (no-op for 64/64 and once built)."""
        key = (int(fx), int(fe))
        if key not in self._prepared:
            _check(load().uds_network_prepare(self.ptr, key[0], key[1]), 'uds_network_prepare')
            self._prepared.add(key)

    def plan_info(self):
        """Tile plan of the fused kernel: dict(fused, node_tiles, link_tiles, p_cap, q_cap, meta_cap, lds_bytes, ...)."""
        info = np.zeros(8, dtype=np.int32)
        _check(load().uds_network_plan_info(self._h, info.ctypes.data), 'uds_network_plan_info')
        keys = ('fused', 'node_tiles', 'link_tiles', 'p_cap', 'q_cap', 'meta_cap', 'lds_bytes', 't_code')
        out = dict(zip(keys, (int(v) for v in info)))
        out['t_node'], out['t_link'] = out['t_code'] // 1000, out['t_code'] % 1000
        return out

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h is not None and _lib is not None:
            _lib.uds_network_destroy(h)


def tile_plan(graph, t_node=48, t_link=48, p_limit=0, q_limit=0, blocks=False):
    """Host-only tile plan of a DrainageGraph (no GPU needed): returns (hdr (T,8) int32, pool int32, caps).
    hdr columns: n_own, n_prim, n_sec, n_inc, n_adj, pool_off, side, meta_len (csrc/tile_plan.hpp).
    blocks=True adds caps['blocks']: the (T, stride) int32 tile blocks the fused d = 64 kernel fetches (fixed-width index
    lists; include/uds_hip.h: uds_tile_plan_blocks), or None when a tile exceeds the byte-wide local indices."""
    lib = load()
    arrs = []
    for c in (graph.adj, graph.edge_adj, graph.inc_n, graph.inc_e):
        arrs += [np.ascontiguousarray(c.rowptr, dtype=np.int32), np.ascontiguousarray(c.col, dtype=np.int32)]
    h = _c_ptr()
    _check(lib.uds_tile_plan_create(*[a.ctypes.data for a in arrs], graph.n_node, graph.n_edge, t_node, t_link,
                                    p_limit, q_limit, ctypes.byref(h)), 'uds_tile_plan_create')
    try:
        nt, pl = _c_i64(), _c_i64()
        caps = np.zeros(3, dtype=np.int32)
        _check(lib.uds_tile_plan_sizes(h, ctypes.byref(nt), ctypes.byref(pl), caps.ctypes.data), 'uds_tile_plan_sizes')
        hdr = np.zeros((nt.value, 8), dtype=np.int32)
        pool = np.zeros(pl.value, dtype=np.int32)
        _check(lib.uds_tile_plan_copy(h, hdr.ctypes.data, pool.ctypes.data), 'uds_tile_plan_copy')
        blk = None
        if blocks:
            stride = _c_i64()
            if lib.uds_tile_plan_blocks(h, None, ctypes.byref(stride)) == 0:
                blk = np.zeros((nt.value, stride.value), dtype=np.int32)
                _check(lib.uds_tile_plan_blocks(h, blk.ctypes.data, ctypes.byref(stride)), 'uds_tile_plan_blocks')
    finally:
        lib.uds_tile_plan_destroy(h)
    out = dict(p_cap=int(caps[0]), q_cap=int(caps[1]), meta_cap=int(caps[2]))
    if blocks:
        out['blocks'] = blk
    return hdr, pool, out


def dense_act(xa, kernel, bias=None, act='linear', xb=None, attn=None):
    """act([xa | xb] @ kernel + bias) on the last axis.  attn=(a_self, a_nbr) also returns the
    GAT attention scalars of the pre-activation rows."""
    lib = load()
    fa = xa.shape[-1]
    fb = 0 if xb is None else xb.shape[-1]
    rows = xa.numel() // fa
    fo = kernel.shape[-1]
    if kernel.shape[0] != fa + fb:
        raise UdsError('kernel has %d input features, inputs give %d' % (kernel.shape[0], fa + fb))
    if xb is not None and xb.shape[:-1] != xa.shape[:-1]:
        raise UdsError('xa %r and xb %r disagree on leading dims' % (tuple(xa.shape), tuple(xb.shape)))
    out = torch.empty(xa.shape[:-1] + (fo,), device=xa.device, dtype=torch.float32)
    s_self = s_nbr = None
    if attn is not None:
        s_self = torch.empty(xa.shape[:-1], device=xa.device, dtype=torch.float32)
        s_nbr = torch.empty_like(s_self)
    if rows == 0:                       # nothing to launch (an empty tensor has no device pointer)
        _dev(xa, 'xa')
        return (out, s_self, s_nbr) if attn is not None else out
    _check(lib.uds_dense_act(_dev(xa, 'xa'), fa, _dev(xb, 'xb', True), fb, rows, _dev(kernel, 'kernel'),
                             _dev(bias, 'bias', True), fo, ACT[act],
                             _dev(attn[0], 'a_self') if attn else None, _dev(attn[1], 'a_nbr') if attn else None,
                             _dev(out, 'out'), _dev(s_self, 's_self', True), _dev(s_nbr, 's_nbr', True), _stream()),
           'uds_dense_act')
    return (out, s_self, s_nbr) if attn is not None else out


def conv1d_causal(x, kernel, bias=None, dilation=1, act='linear'):
    """keras Conv1D(padding='causal', dilation_rate) along axis 1 of x (B,T,R,F), no transposes; kernel (taps,F,H)."""
    lib = load()
    if x.dim() != 4 or kernel.dim() != 3 or kernel.shape[1] != x.shape[-1]:
        raise UdsError('conv1d_causal: x %r / kernel %r' % (tuple(x.shape), tuple(kernel.shape)))
    B, T, R, F = x.shape
    taps, _, H = kernel.shape
    out = torch.empty((B, T, R, H), device=x.device, dtype=torch.float32)
    if out.numel() == 0:
        _dev(x, 'x')
        return out
    _check(lib.uds_conv1d_causal(_dev(x, 'x'), B, T, R, F, _dev(kernel, 'kernel'), _dev(bias, 'bias', True), taps, dilation, H,
                                 ACT[act], _dev(out, 'out'), _stream()), 'uds_conv1d_causal')
    return out


def recurrent_fused_supported(F, kind):
    """Input forms uds_recurrent_fused takes: F = 64 / 128 input rows (W + U must fit the LDS), F = 0 = given projection."""
    return bool(load().uds_recurrent_fused_supported(int(F), 0 if kind == 'GRU' else 1))


def recurrent_pack(kernel, recurrent_kernel):
    """MFMA fragments of a GRU / LSTM layer with 64 units: the G 64-column slices of `kernel` (F, G*64) -- none when kernel is
    None (the input projection is computed separately) --, then of `recurrent_kernel` (64, G*64)."""
    G = recurrent_kernel.shape[1] // 64
    if tuple(recurrent_kernel.shape) != (64, G * 64) or G not in (3, 4) or (kernel is not None and kernel.shape[1] != G * 64):
        raise UdsError('recurrent_pack: kernel %r / recurrent kernel %r (F x G*64 and 64 x G*64 with G = 3 or 4)'
                       % (None if kernel is None else tuple(kernel.shape), tuple(recurrent_kernel.shape)))
    mats = [recurrent_kernel] if kernel is None else [kernel, recurrent_kernel]
    return torch.cat([rowgemm_pack(m[:, 64 * g:64 * (g + 1)].contiguous()) for m in mats for g in range(G)]).contiguous()


def recurrent_fused(x, packed, b_in, b_rec, kind, projected=False):
    """Hidden states (B, T, R, 64) of a GRU / LSTM layer with 64 units in one launch (uds_recurrent_fused).  projected=True:
    x is the input projection (B, T, R, G*64) incl. its bias."""
    lib = load()
    B, T, R, F = x.shape
    out = torch.empty((B, T, R, 64), device=x.device, dtype=torch.float32)
    if out.numel():
        _check(lib.uds_recurrent_fused(_dev(x, 'x'), 0 if projected else F, packed.data_ptr(), _dev(b_in, 'b_in', True), _dev(b_rec, 'b_rec', True),
                                       B, T, R, 0 if kind == 'GRU' else 1, _dev(out, 'out'), _stream()), 'uds_recurrent_fused')
    return out


def attn_sum_pool(x, attn_kernel):
    """GlobalAttnSumPool: x (B, R, F), attn_kernel (F,) or (F, 1) -> (B, F) (uds_attn_sum_pool)."""
    lib = load()
    B, R, F = x.shape
    k = attn_kernel.reshape(-1).contiguous()
    if k.numel() != F:
        raise UdsError('attn_sum_pool: x %r, attn_kernel %r' % (tuple(x.shape), tuple(attn_kernel.shape)))
    out = torch.empty((B, F), device=x.device, dtype=torch.float32)
    if B:
        _check(lib.uds_attn_sum_pool(_dev(x, 'x'), _dev(k, 'attn_kernel'), B, R, F, _dev(out, 'out'), _stream()), 'uds_attn_sum_pool')
    return out


def dropout(x, rate, seed, offset):
    """keras Dropout in training mode on a contiguous fp32 device tensor (uds_dropout): kept elements scaled by 1 / (1 - rate);
    the mask depends on (seed, offset + flat index) only."""
    lib = load()
    x = x.contiguous()
    out = torch.empty_like(x)
    if out.numel():
        _check(lib.uds_dropout(_dev(x, 'x'), x.numel(), float(rate), int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset) & 0xFFFFFFFFFFFFFFFF,
                               _dev(out, 'out'), _stream()), 'uds_dropout')
    return out


def recurrent_forward_train(xp, recurrent_kernel, recurrent_bias, kind):
    """(h, c): hidden states (B, T, R, H) as recurrent_forward, plus the LSTM's cell states (None for the GRU) -- what
    recurrent_backward needs from the forward pass."""
    lib = load()
    B, T, R, GH = xp.shape
    G = {'GRU': 3, 'LSTM': 4}[kind]
    H = GH // G
    if GH != G * H or tuple(recurrent_kernel.shape) != (H, GH):
        raise UdsError('recurrent_forward: xp %r, recurrent kernel %r for a %s' % (tuple(xp.shape), tuple(recurrent_kernel.shape), kind))
    out = torch.empty((B, T, R, H), device=xp.device, dtype=torch.float32)
    c = torch.empty_like(out) if kind == 'LSTM' else None
    if out.numel():
        _check(lib.uds_recurrent_forward_train(_dev(xp, 'xp'), _dev(recurrent_kernel, 'recurrent_kernel'), _dev(recurrent_bias, 'recurrent_bias', True),
                                               B, T, R, H, 0 if kind == 'GRU' else 1, _dev(out, 'out'), _dev(c, 'c', True), _stream()),
               'uds_recurrent_forward_train')
    return out, c


def recurrent_pack_bwd(recurrent_kernel):
    """MFMA fragments for uds_recurrent_backward: the G 64 x 64 slices of U, then of U_g^T (64 units only)."""
    G = recurrent_kernel.shape[1] // 64
    if tuple(recurrent_kernel.shape) != (64, G * 64) or G not in (3, 4):
        raise UdsError('recurrent_pack_bwd: recurrent kernel %r (64 x G*64 with G = 3 or 4)' % (tuple(recurrent_kernel.shape),))
    sl = [recurrent_kernel[:, 64 * g:64 * (g + 1)] for g in range(G)]
    return torch.cat([rowgemm_pack(m.contiguous()) for m in sl] + [rowgemm_pack(m.t().contiguous()) for m in sl]).contiguous()


def recurrent_backward(xp, packed, recurrent_bias, h, c, gh, kind):
    """(dxp (B, T, R, G*64), darec (G, B, T, R, 64)) of a 64-unit GRU / LSTM layer: back-propagation through time in one launch."""
    lib = load()
    B, T, R, GH = xp.shape
    G = {'GRU': 3, 'LSTM': 4}[kind]
    if GH != G * 64 or tuple(h.shape) != (B, T, R, 64) or tuple(gh.shape) != (B, T, R, 64):
        raise UdsError('recurrent_backward: xp %r, h %r, gh %r for a 64-unit %s' % (tuple(xp.shape), tuple(h.shape), tuple(gh.shape), kind))
    dxp = torch.empty_like(xp)
    darec = torch.empty((G, B, T, R, 64), device=xp.device, dtype=torch.float32)
    if dxp.numel():
        _check(lib.uds_recurrent_backward(_dev(xp, 'xp'), packed.data_ptr(), _dev(recurrent_bias, 'recurrent_bias', True), _dev(h, 'h'),
                                          _dev(c, 'c', True), _dev(gh, 'gh'), B, T, R, 0 if kind == 'GRU' else 1, _dev(dxp, 'dxp'),
                                          _dev(darec, 'darec'), _stream()), 'uds_recurrent_backward')
    return dxp, darec


def recurrent_forward(xp, recurrent_kernel, recurrent_bias, kind):
    """Hidden states (B, T, R, H) of a GRU (kind 'GRU') or LSTM from the input projections xp (B, T, R, G*H)."""
    lib = load()
    B, T, R, GH = xp.shape
    G = {'GRU': 3, 'LSTM': 4}[kind]
    H = GH // G
    if GH != G * H or tuple(recurrent_kernel.shape) != (H, GH):
        raise UdsError('recurrent_forward: xp %r, recurrent kernel %r for a %s' % (tuple(xp.shape), tuple(recurrent_kernel.shape), kind))
    out = torch.empty((B, T, R, H), device=xp.device, dtype=torch.float32)
    if out.numel():
        _check(lib.uds_recurrent_forward(_dev(xp, 'xp'), _dev(recurrent_kernel, 'recurrent_kernel'), _dev(recurrent_bias, 'recurrent_bias', True),
                                         B, T, R, H, 0 if kind == 'GRU' else 1, _dev(out, 'out'), _stream()), 'uds_recurrent_forward')
    return out


def rowgemm_supported(k_total, f_in, f_out):
    """Shapes the matrix-core row GEMM takes: input width a multiple of 32, at most 64 outputs, and the packed weights
    (K/32 * ceil(f_out/16) * 2 KiB) next to at least 32 KiB of per-wave DMA rings and the store tiles inside the 160 KiB LDS."""
    mb = 1 if f_out <= 16 else 2 if f_out <= 32 else 4
    return (f_in % 32 == 0 and k_total % 32 == 0 and 0 < f_out <= 64
            and (k_total // 32) * mb * 2048 + 256 + 8 * 2 * 2048 + 8 * 16 * (36 if mb >= 2 else 20) * 4 <= 160 * 1024)


def rowgemm_pack(kernel2d):
    """Pre-split a (K, f_out) kernel into bf16 hi/lo MFMA fragments (once per parameter update)."""
    lib = load()
    K, fo = kernel2d.shape
    nbytes = lib.uds_rowgemm_packed_bytes(K, fo)
    if nbytes <= 0:
        raise UdsError('rowgemm_pack: unsupported shape %r' % ((K, fo),))
    out = torch.empty(nbytes // 4, device=kernel2d.device, dtype=torch.float32)
    _check(lib.uds_rowgemm_pack(_dev(kernel2d, 'kernel'), K, fo, out.data_ptr(), _stream()), 'uds_rowgemm_pack')
    return out


def diffusion_forward(csr, vals, c0, r, tot, act='tanh'):
    """out[s, i, q] = act(c0[q] tot[s] + sum_p vals[p, q] r[s, col[p]]): DiffusionConv on the CSR support (uds_diffusion_forward)."""
    lib = load()
    S, C = r.shape[0], vals.shape[1]
    if tuple(vals.shape) != (csr.nnz, C) or tuple(c0.shape) != (C,) or tuple(r.shape) != (S, csr.n_cols) or tuple(tot.shape) != (S,):
        raise UdsError('diffusion_forward: vals %r c0 %r r %r tot %r for a %d x %d pattern with %d entries'
                       % (tuple(vals.shape), tuple(c0.shape), tuple(r.shape), tuple(tot.shape), csr.n_rows, csr.n_cols, csr.nnz))
    out = torch.empty((S, csr.n_rows, C), device=r.device, dtype=torch.float32)
    if out.numel():
        _check(lib.uds_diffusion_forward(csr.ptr, _dev(vals, 'vals'), _dev(c0, 'c0'), _dev(r, 'r'), _dev(tot, 'tot'), S, C, ACT[act],
                                         _dev(out, 'out'), _stream()), 'uds_diffusion_forward')
    return out


def halo_pack(x, e, idx_x, idx_e):
    """One message buffer (S, nx + ne, F) = [x[:, idx_x] | e[:, idx_e]] (int32 device index tensors): one launch per peer."""
    lib = load()
    S, n_x, F = x.shape
    n_e = e.shape[1]
    nx, ne = int(idx_x.numel()), int(idx_e.numel())
    buf = torch.empty((S, nx + ne, F), device=x.device, dtype=torch.float32)
    if buf.numel():
        _check(lib.uds_halo_pack(_dev(x, 'x'), n_x, _dev(e, 'e'), n_e, S, F, _dev_i32(idx_x, 'idx_x') if nx else None, nx,
                                 _dev_i32(idx_e, 'idx_e') if ne else None, ne, _dev(buf, 'buf'), _stream()), 'uds_halo_pack')
    return buf


def halo_unpack(buf, x, e, idx_x, idx_e):
    """x[:, idx_x], e[:, idx_e] = the two parts of a received message buffer (in place)."""
    lib = load()
    S, n_x, F = x.shape
    nx, ne = int(idx_x.numel()), int(idx_e.numel())
    if tuple(buf.shape) != (S, nx + ne, F):
        raise UdsError('halo_unpack: buffer %r for %d + %d rows of %r' % (tuple(buf.shape), nx, ne, tuple(x.shape)))
    if buf.numel():
        _check(lib.uds_halo_unpack(_dev(buf, 'buf'), S, F, _dev_i32(idx_x, 'idx_x') if nx else None, nx,
                                   _dev_i32(idx_e, 'idx_e') if ne else None, ne, _dev(x, 'x'), n_x, _dev(e, 'e'), e.shape[1], _stream()),
               'uds_halo_unpack')


def remainder_pack(rest):
    """Pre-split the dense off-support part `rest` (R, M) of a trained NodeEdge into bf16 hi/lo planes (once per update)."""
    lib = load()
    if rest.dim() != 2 or rest.numel() == 0:
        raise UdsError('remainder_pack: rest %r' % (tuple(rest.shape),))
    R, M = rest.shape
    out = torch.empty(lib.uds_remainder_packed_bytes(R, M) // 4, device=rest.device, dtype=torch.float32)
    _check(lib.uds_remainder_pack(_dev(rest, 'rest'), R, M, out.data_ptr(), _stream()), 'uds_remainder_pack')
    return out


def remainder_forward(packed, shape, x):
    """out[..., r, :] = sum_m rest[r, m] x[..., m, :] on the matrix cores (split-bf16): `packed` = remainder_pack(rest),
    shape = rest.shape, x (..., M, h) with h % 4 == 0, h <= 64."""
    lib = load()
    R, M = shape
    if x.dim() < 2 or x.shape[-2] != M:
        raise UdsError('remainder_forward: x %r against rest (%d, %d)' % (tuple(x.shape), R, M))
    h = x.shape[-1]
    S = x.numel() // (M * h) if M * h else 0
    out = torch.empty(tuple(x.shape[:-2]) + (R, h), device=x.device, dtype=torch.float32)
    if S == 0:
        _dev(x, 'x')
        return out
    ws = torch.empty(lib.uds_remainder_workspace_bytes(R, M, S, h) // 4, device=x.device, dtype=torch.float32)
    _check(lib.uds_remainder_forward(packed.data_ptr(), R, M, _dev(x, 'x'), S, h, ws.data_ptr(), _dev(out, 'out'), _stream()),
           'uds_remainder_forward')
    return out


def remainder_forward_dense(packed, shape, e, packed_w, bias, act, h):
    """rest @ act(e W + b) without the fp32 x_e in between (uds_remainder_forward_dense): e (..., M, F) with F 64 or 128, packed_w =
    rowgemm_pack(W (F, h)), h 32 or 64 -> (..., R, h)."""
    lib = load()
    R, M = shape
    if e.dim() < 2 or e.shape[-2] != M:
        raise UdsError('remainder_forward_dense: e %r against rest (%d, %d)' % (tuple(e.shape), R, M))
    F = e.shape[-1]
    S = e.numel() // (M * F) if M * F else 0
    out = torch.empty(tuple(e.shape[:-2]) + (R, h), device=e.device, dtype=torch.float32)
    if S == 0:
        _dev(e, 'e')
        return out
    ws = torch.empty(lib.uds_remainder_workspace_bytes(R, M, S, h) // 4, device=e.device, dtype=torch.float32)
    _check(lib.uds_remainder_forward_dense(packed.data_ptr(), R, M, _dev(e, 'e'), F, packed_w.data_ptr(), _dev(bias, 'bias', True), ACT[act], S, h,
                                           ws.data_ptr(), _dev(out, 'out'), _stream()), 'uds_remainder_forward_dense')
    return out


def rowgemm_forward(x, packed, bias, f_out, act='linear', taps=1, dilation=1):
    """Matrix-core Dense (taps=1, any leading dims) or causal Conv1D (x (B,T,R,F), taps = kernel size)."""
    lib = load()
    F = x.shape[-1]
    if taps == 1:
        B, T, R = 1, 1, x.numel() // F
        out_shape = tuple(x.shape[:-1]) + (f_out,)
    else:
        if x.dim() != 4:
            raise UdsError('conv needs x (B,T,R,F), got %r' % (tuple(x.shape),))
        B, T, R = x.shape[:3]
        out_shape = (B, T, R, f_out)
    out = torch.empty(out_shape, device=x.device, dtype=torch.float32)
    if out.numel() == 0:
        _dev(x, 'x')
        return out
    _check(lib.uds_rowgemm_forward(_dev(x, 'x'), B, T, R, F, packed.data_ptr(), _dev(bias, 'bias', True), taps, dilation, f_out,
                                   ACT[act], _dev(out, 'out'), _stream()), 'uds_rowgemm_forward')
    return out


def rowgemm_forward_pair(x0, packed0, bias0, x1, packed1, bias1, f_out, act='linear', taps=1, dilation=1):
    """Two causal Conv1D / Dense problems of one layer shape -- x0 (B,T,R0,F), x1 (B,T,R1,F) -- in one launch when both are small
    (uds_rowgemm_forward_pair).  Returns (out0, out1)."""
    lib = load()
    B, T, R0, F = x0.shape
    R1 = x1.shape[2]
    if tuple(x1.shape) != (B, T, R1, F):
        raise UdsError('rowgemm_forward_pair: x0 %r and x1 %r differ in more than the row count' % (tuple(x0.shape), tuple(x1.shape)))
    out0 = torch.empty((B, T, R0, f_out), device=x0.device, dtype=torch.float32)
    out1 = torch.empty((B, T, R1, f_out), device=x0.device, dtype=torch.float32)
    if out0.numel() and out1.numel():
        _check(lib.uds_rowgemm_forward_pair(_dev(x0, 'x0'), R0, packed0.data_ptr(), _dev(bias0, 'bias0', True), _dev(out0, 'out0'), _dev(x1, 'x1'), R1,
                                            packed1.data_ptr(), _dev(bias1, 'bias1', True), _dev(out1, 'out1'), B, T, F, taps, dilation, f_out,
                                            ACT[act], _stream()), 'uds_rowgemm_forward_pair')
    return out0, out1


def rowgemm_cat(x, x2, packed, bias, f_out, act='linear', out=None, col0=0):
    """Matrix-core Dense on rows [x | x2] (x2 None: x alone) written to columns [col0, col0 + f_out) of `out`
    (..., ldo) -- allocated (..., f_out) when None.  No concatenation copies (uds_rowgemm_forward_cat)."""
    lib = load()
    F1 = x.shape[-1]
    F2 = 0 if x2 is None else x2.shape[-1]
    rows = x.numel() // F1
    if x2 is not None and x2.numel() // F2 != rows:
        raise UdsError('rowgemm_cat: x %r and x2 %r have different row counts' % (tuple(x.shape), tuple(x2.shape)))
    if out is None:
        out = torch.empty(tuple(x.shape[:-1]) + (f_out,), device=x.device, dtype=torch.float32)
    ldo = out.shape[-1]
    if out.numel() // ldo != rows:
        raise UdsError('rowgemm_cat: out %r does not have %d rows' % (tuple(out.shape), rows))
    if rows == 0:
        _dev(x, 'x')
        return out
    _check(lib.uds_rowgemm_forward_cat(_dev(x, 'x'), F1, _dev(x2, 'x2', True), F2, 1, 1, rows, packed.data_ptr(), _dev(bias, 'bias', True),
                                       1, 1, f_out, ACT[act], _dev(out, 'out'), ldo, col0, _stream()), 'uds_rowgemm_forward_cat')
    return out


def dense_cumsum(x, packed, bias=None, res=None, act='linear'):
    """act(cumsum over axis 1 of (x @ kernel + bias) + res): x (B,T,R,64), kernel (64,64) packed by rowgemm_pack,
    res (B,1,R,64) -- Dense + prefix sum + residual + activation in one kernel (uds_dense_cumsum)."""
    lib = load()
    B, T, R, F = x.shape
    if F != 64:
        raise UdsError('dense_cumsum: 64 -> 64 only, got %d inputs' % F)
    if res is not None and tuple(res.shape) != (B, 1, R, 64):
        raise UdsError('dense_cumsum: res must be %r, got %r' % ((B, 1, R, 64), tuple(res.shape)))
    out = torch.empty((B, T, R, 64), device=x.device, dtype=torch.float32)
    if out.numel() == 0:
        _dev(x, 'x')
        return out
    _check(lib.uds_dense_cumsum(_dev(x, 'x'), B, T, R, packed.data_ptr(), _dev(bias, 'bias', True), _dev(res, 'res', True), ACT[act],
                                _dev(out, 'out'), _stream()), 'uds_dense_cumsum')
    return out


class _Heads(ctypes.Structure):
    _fields_ = [('a_packed', ctypes.c_void_p), ('a_bias', ctypes.c_void_p), ('h_packed', ctypes.c_void_p * 5), ('h_bias', ctypes.c_void_p * 5),
                ('f_packed', ctypes.c_void_p), ('f_bias', ctypes.c_void_p), ('n_a', ctypes.c_int32), ('act_a', ctypes.c_int32),
                ('n_hidden', ctypes.c_int32), ('act_h', ctypes.c_int32), ('act_f', ctypes.c_int32)]


def dense_cumsum_heads(x, packed, bias, res, act, head_a, hidden=(), head_f=None):
    """uds_dense_cumsum_heads: act(cumsum_t(x @ kernel + bias) + res) consumed by its heads without being written.
    head_a = (packed (64, n_a), bias, n_a, activation); hidden = [(packed, bias), ...] the Dense(32) layers of the second
    head with their common activation in hidden_act = head_f[3]; head_f = (packed (32, 1), bias, activation, hidden activation).
    Returns (B, T, R, n_a + (1 if hidden else 0))."""
    lib = load()
    B, T, R, F = x.shape
    if F != 64:
        raise UdsError('dense_cumsum_heads: 64 -> 64 only, got %d inputs' % F)
    if res is not None and tuple(res.shape) != (B, 1, R, 64):
        raise UdsError('dense_cumsum_heads: res must be %r, got %r' % ((B, 1, R, 64), tuple(res.shape)))
    hd = _Heads()
    keep = [head_a[0], head_a[1]]
    hd.a_packed, hd.a_bias, hd.n_a, hd.act_a = head_a[0].data_ptr(), _dev(head_a[1], 'a_bias', True), int(head_a[2]), ACT[head_a[3]]
    hd.n_hidden = len(hidden)
    for i, (pk, bs) in enumerate(hidden):
        hd.h_packed[i], hd.h_bias[i] = pk.data_ptr(), _dev(bs, 'h_bias', True)
        keep += [pk, bs]
    if hidden:
        hd.f_packed, hd.f_bias, hd.act_f, hd.act_h = head_f[0].data_ptr(), _dev(head_f[1], 'f_bias', True), ACT[head_f[2]], ACT[head_f[3]]
        keep += [head_f[0], head_f[1]]
    out = torch.empty((B, T, R, hd.n_a + (1 if hidden else 0)), device=x.device, dtype=torch.float32)
    if out.numel():
        _check(lib.uds_dense_cumsum_heads(_dev(x, 'x'), B, T, R, packed.data_ptr(), _dev(bias, 'bias', True), _dev(res, 'res', True), ACT[act],
                                          ctypes.byref(hd), _dev(out, 'out'), _stream()), 'uds_dense_cumsum_heads')
    return out


def cumsum_act(x, res=None, act='linear'):
    """act(cumsum over axis 1 of x (B,T,R,F) + res (B,1,R,F))."""
    lib = load()
    B, T, R, F = x.shape
    if res is not None and tuple(res.shape) != (B, 1, R, F):
        raise UdsError('cumsum_act: res must be %r, got %r' % ((B, 1, R, F), tuple(res.shape)))
    out = torch.empty_like(x)
    if out.numel() == 0:
        _dev(x, 'x')
        return out
    _check(lib.uds_cumsum_act(_dev(x, 'x'), _dev(res, 'res', True), B, T, R, F, ACT[act], _dev(out, 'out'), _stream()),
           'uds_cumsum_act')
    return out


def flow_balance(handle, sign, flow, scale_in, scale_out):
    """Link -> node flow balance (emulator.py:717-724): flow (S,E) -> q_in, q_out (S,N)."""
    lib = load()
    if flow.dim() != 2 or flow.shape[1] != handle.n_cols:
        raise UdsError('flow must be (S,%d), got %r' % (handle.n_cols, tuple(flow.shape)))
    S = flow.shape[0]
    q_in = torch.empty((S, handle.n_rows), device=flow.device, dtype=torch.float32)
    q_out = torch.empty_like(q_in)
    if q_in.numel() == 0:
        return q_in, q_out
    _check(lib.uds_flow_balance(handle.ptr, _dev(sign, 'sign'), _dev(flow, 'flow'), S, _dev(scale_in, 'scale_in'),
                                _dev(scale_out, 'scale_out'), _dev(q_in, 'q_in'), _dev(q_out, 'q_out'), _stream()),
           'uds_flow_balance')
    return q_in, q_out


def roll_update(handle, sign, span_e, mini_e, scale_in, scale_out, y, ey, b, x, ex, flood):
    """The post-forward part of one autoregressive chunk (include/uds_hip.h: uds_roll_update): returns preds (B,so,N,cy+2) and
    shifts the state windows x (B,T,N,cy+3), ex (B,T,E,ce+1) IN PLACE, feeding the prediction back."""
    lib = load()
    B, so, N, cy = y.shape
    ce, T = ey.shape[-1], x.shape[1]
    if (tuple(ey.shape[:3]) != (B, so, handle.n_cols) or tuple(b.shape) != (B, so, N, 1) or tuple(x.shape) != (B, T, N, cy + 3) or
            tuple(ex.shape) != (B, T, handle.n_cols, ce + 1) or N != handle.n_rows):
        raise UdsError('roll_update: inconsistent shapes y %r ey %r b %r x %r ex %r' % tuple(tuple(t.shape) for t in (y, ey, b, x, ex)))
    if not (x.is_contiguous() and ex.is_contiguous()):
        raise UdsError('roll_update: the state windows are updated in place and must be contiguous')
    preds = torch.empty((B, so, N, cy + 2), device=y.device, dtype=torch.float32)
    _check(lib.uds_roll_update(handle.ptr, _dev(sign, 'sign'), _dev(span_e, 'span_e'), _dev(mini_e, 'mini_e'), _dev(scale_in, 'scale_in'),
                               _dev(scale_out, 'scale_out'), _dev(y.contiguous(), 'y'), cy, _dev(ey.contiguous(), 'ey'), ce, _dev(b.contiguous(), 'b'),
                               B, so, T, int(bool(flood)), _dev(x, 'x'), _dev(ex, 'ex'), _dev(preds, 'preds'), _stream()), 'uds_roll_update')
    return preds


def csr_spmm(handle, val, x, bias=None, act='linear'):
    """out[s,r,:] = act(sum_p val[p] x[s,col[p],:] + bias); x:(S,n_cols,F) -> (S,n_rows,F)."""
    lib = load()
    if x.dim() != 3 or x.shape[1] != handle.n_cols:
        raise UdsError('x must be (S,%d,F), got %r' % (handle.n_cols, tuple(x.shape)))
    if val is not None and val.numel() != handle.nnz:
        raise UdsError('val has %d entries, pattern has %d' % (val.numel(), handle.nnz))
    S, _, F = x.shape
    out = torch.empty((S, handle.n_rows, F), device=x.device, dtype=torch.float32)
    if out.numel() == 0:
        _dev(x, 'x')
        return out
    _check(lib.uds_csr_spmm(handle.ptr, _dev(val, 'val', True), _dev(x, 'x'), S, F, _dev(bias, 'bias', True), ACT[act],
                            _dev(out, 'out'), _stream()), 'uds_csr_spmm')
    return out


def gat_forward(handle, xa, kernel, a_self, a_nbr, bias=None, act='relu', xb=None, return_workspace=False):
    """Single-head GATConv over a CSR pattern with self loops; xa:(S,n,fa) [| xb:(S,n,fb)] -> (S,n,d).
    return_workspace=True also returns (hx (S,n,d), s_self (S,n), s_nbr (S,n)) -- views of the workspace, for the
    backward pass."""
    lib = load()
    if xa.dim() != 3 or xa.shape[1] != handle.n_rows:
        raise UdsError('x must be (S,%d,F), got %r' % (handle.n_rows, tuple(xa.shape)))
    S, n, fa = xa.shape
    fb = 0 if xb is None else xb.shape[-1]
    d = kernel.shape[-1]
    if kernel.numel() != (fa + fb) * d:
        raise UdsError('kernel %r does not match %d input features' % (tuple(kernel.shape), fa + fb))
    out = torch.empty((S, n, d), device=xa.device, dtype=torch.float32)
    if out.numel() == 0:
        _dev(xa, 'xa')
        return out
    ws = torch.empty(lib.uds_gat_workspace_floats(n, S, d), device=xa.device, dtype=torch.float32)
    _check(lib.uds_gat_forward(handle.ptr, _dev(xa, 'xa'), fa, _dev(xb, 'xb', True), fb, S, _dev(kernel, 'kernel'),
                               _dev(a_self, 'a_self'), _dev(a_nbr, 'a_nbr'), _dev(bias, 'bias', True), d, ACT[act],
                               _dev(ws, 'workspace'), _dev(out, 'out'), _stream()), 'uds_gat_forward')
    if return_workspace:
        m = S * n
        return out, (ws[:m * d].view(S, n, d), ws[m * d:m * d + m].view(S, n), ws[m * d + m:m * d + 2 * m].view(S, n))
    return out


def gat_aggregate(handle, hx, s_self, s_nbr, bias=None, act='relu', edge_mask=None, coef=None):
    """Attention softmax + neighbour sum of a GATConv from precomputed hx (S,n,d), s_self / s_nbr (S,n).
    edge_mask (S, nnz): per-snapshot 0/1 over the pattern's entries (`use_adj`; the diagonal always takes part).
    coef (S, nnz): multiplier of the normalised coefficients (Spektral's attention dropout in training, uds_gat_aggregate_coef)."""
    lib = load()
    S, n, d = hx.shape
    if coef is not None:
        if edge_mask is not None:
            raise UdsError('gat_aggregate: edge_mask and coef together are not built')
        if tuple(coef.shape) != (S, handle.nnz) or n != handle.n_rows or tuple(s_self.shape) != (S, n) or tuple(s_nbr.shape) != (S, n):
            raise UdsError('gat_aggregate: coef %r / hx %r for %d snapshots of a %d-row, %d-entry pattern' %
                           (tuple(coef.shape), tuple(hx.shape), S, handle.n_rows, handle.nnz))
        out = torch.empty_like(hx)
        if out.numel():
            _check(lib.uds_gat_aggregate_coef(handle.ptr, _dev(hx, 'hx'), _dev(s_self, 's_self'), _dev(s_nbr, 's_nbr'), _dev(bias, 'bias', True),
                                              _dev(coef, 'coef'), S, d, ACT[act], _dev(out, 'out'), _stream()), 'uds_gat_aggregate_coef')
        return out
    if edge_mask is not None:
        if tuple(edge_mask.shape) != (S, handle.nnz):
            raise UdsError('gat_aggregate: edge_mask %r for %d snapshots of a %d-entry pattern' % (tuple(edge_mask.shape), S, handle.nnz))
        if n != handle.n_rows or tuple(s_self.shape) != (S, n) or tuple(s_nbr.shape) != (S, n):
            raise UdsError('gat_aggregate: hx %r does not match a %d-row pattern' % (tuple(hx.shape), handle.n_rows))
        out = torch.empty_like(hx)
        if out.numel():
            _check(lib.uds_gat_aggregate_masked(handle.ptr, _dev(hx, 'hx'), _dev(s_self, 's_self'), _dev(s_nbr, 's_nbr'), _dev(bias, 'bias', True),
                                                _dev(edge_mask, 'edge_mask'), S, d, ACT[act], _dev(out, 'out'), _stream()),
                   'uds_gat_aggregate_masked')
        return out
    if n != handle.n_rows or tuple(s_self.shape) != (S, n) or tuple(s_nbr.shape) != (S, n):
        raise UdsError('gat_aggregate: hx %r, s_self %r, s_nbr %r do not match a %d-row pattern' %
                       (tuple(hx.shape), tuple(s_self.shape), tuple(s_nbr.shape), handle.n_rows))
    out = torch.empty_like(hx)
    if out.numel() == 0:
        _dev(hx, 'hx')
        return out
    _check(lib.uds_gat_aggregate(handle.ptr, _dev(hx, 'hx'), _dev(s_self, 's_self'), _dev(s_nbr, 's_nbr'), _dev(bias, 'bias', True), S, d,
                                 ACT[act], _dev(out, 'out'), _stream()), 'uds_gat_aggregate')
    return out


def gat_backward(handle, handle_t, perm_t, grad, hx, s_self, s_nbr, a_self, a_nbr, coef=None):
    """Reverse mode of the attention / aggregation part of gat_forward (uds_gat_backward): grad = dL/d(pre-activation)
    (S,n,d) -> d_hx (S,n,d), ds_self (S,n), ds_nbr (S,n).  coef: the attention-dropout multiplier of the forward pass, if any."""
    lib = load()
    S, n, d = grad.shape
    d_hx = torch.empty_like(grad)
    ds_self = torch.empty((S, n), device=grad.device, dtype=torch.float32)
    ds_nbr = torch.empty_like(ds_self)
    if grad.numel() == 0:
        _dev(grad, 'grad')
        return d_hx, ds_self, ds_nbr
    ws = torch.empty((2, S, max(handle.nnz, 1)), device=grad.device, dtype=torch.float32)
    _check(lib.uds_gat_backward_coef(handle.ptr, handle_t.ptr, _dev_i32(perm_t, 'perm_t'), _dev(grad, 'grad'), _dev(hx, 'hx'),
                                     _dev(s_self, 's_self'), _dev(s_nbr, 's_nbr'), _dev(a_self, 'a_self'), _dev(a_nbr, 'a_nbr'),
                                     _dev(coef, 'coef', True), S, d, _dev(ws[0], 'alpha_ws'), _dev(ws[1], 'de_ws'), _dev(d_hx, 'd_hx'),
                                     _dev(ds_self, 'ds_self'), _dev(ds_nbr, 'ds_nbr'), _stream()), 'uds_gat_backward_coef')
    return d_hx, ds_self, ds_nbr


def wgrad_supported(F, H, with_bias=True):
    return load().uds_wgrad_workspace_floats(128, F, H, int(with_bias)) > 0


def wgrad(a, g, shift=0, with_bias=True):
    """d_kernel (F,H) [, d_bias (H)] of out = a @ kernel (+ bias): sums over all rows of a (B,T,R,F)^T g (B,T,R,H), for a
    causal Conv1D tap with the input `shift` time steps back.  Matrix cores, split-bf16, deterministic."""
    lib = load()
    B, T, R, F = a.shape
    H = g.shape[-1]
    if tuple(g.shape[:-1]) != (B, T, R):
        raise UdsError('wgrad: a %r and g %r disagree' % (tuple(a.shape), tuple(g.shape)))
    n_ws = lib.uds_wgrad_workspace_floats(max(B * T * R, 1), F, H, int(with_bias))
    if n_ws <= 0:
        raise UdsError('wgrad: F=%d / H=%d not supported' % (F, H))
    dk = torch.empty((F, H), device=a.device, dtype=torch.float32)
    db = torch.empty(H, device=a.device, dtype=torch.float32) if with_bias else None
    ws = torch.empty(n_ws, device=a.device, dtype=torch.float32)
    _check(lib.uds_wgrad(_dev(a, 'a'), _dev(g, 'g'), B, T, R, F, H, shift, int(with_bias), _dev(ws, 'workspace'), _dev(dk, 'd_kernel'),
                         _dev(db, 'd_bias', True), _stream()), 'uds_wgrad')
    return dk, db


def csr_sddmm(handle, a, b):
    """out[k] = sum_s <a[s,row(k),:], b[s,col(k),:]> per pattern entry; a:(S,n_rows,F), b:(S,n_cols,F) -> (nnz,)."""
    lib = load()
    if a.dim() != 3 or b.dim() != 3 or a.shape[1] != handle.n_rows or b.shape[1] != handle.n_cols or a.shape[0] != b.shape[0] \
            or a.shape[2] != b.shape[2]:
        raise UdsError('sddmm: a must be (S,%d,F) and b (S,%d,F), got %r, %r' % (handle.n_rows, handle.n_cols, tuple(a.shape),
                                                                               tuple(b.shape)))
    out = torch.zeros(handle.nnz, device=a.device, dtype=torch.float32)
    if handle.nnz == 0 or a.numel() == 0:
        _dev(a, 'a')
        return out
    _check(lib.uds_csr_sddmm(handle.ptr, _dev(a, 'a'), _dev(b, 'b'), a.shape[0], a.shape[2], _dev(out, 'out'), _stream()),
           'uds_csr_sddmm')
    return out


def _spatial_params(p):
    sp = SpatialParams()
    for name, _ in SpatialParams._fields_:
        if name == 'packed':
            t = p.get('packed')
            sp.packed = None if t is None else t.data_ptr()
        else:
            setattr(sp, name, _dev(p[name], name, allow_none=name.endswith('_b')))
    return sp


def spatial_pack_weights(p, fx, fe, h, d):
    """Pre-split the four kernels of a spatial layer into the fused kernel's bf16 hi/lo MFMA fragments (done once per
    parameter update instead of in every forward call).  Returns the device buffer to pass as p['packed']."""
    lib = load()
    out = torch.empty(lib.uds_spatial_packed_bytes() // 4, device=p['xe_k'].device, dtype=torch.float32)
    q = dict(p)
    q['packed'] = None
    _check(lib.uds_spatial_pack_weights(ctypes.byref(_spatial_params(q)), fx, fe, h, d, out.data_ptr(), _stream()),
           'uds_spatial_pack_weights')
    return out


def spatial_layer_forward(net, p, x, e, h, d, act='relu', flags=0, xb=None, eb=None, rem_x=None, rem_e=None):
    """One spatial-block loop body (`emulator.py:225-230`).  p: dict of the 14 tensors of
    uds_spatial_params_t.  x:(S,N,fx), e:(S,E,fe) -> (S,N,d), (S,E,d).  flags: FLAG_* of the C ABI.
    xb (S,N,32) / eb (S,E,32): extra columns appended to a 64-wide x / e without materialising the concatenation
    (uds_spatial_layer_forward_split; fused kernel only).
    rem_x (S,N,h) / rem_e (S,E,h): the dense remainder of a trained NodeEdge, added to the aggregates inside the d = 128 fused
    kernel (uds_spatial_layer_forward_rem; fused kernel only)."""
    lib = load()
    S, N, fx = x.shape
    _, E, fe = e.shape
    if e.shape[0] != S or N != net.graph.n_node or E != net.graph.n_edge:
        raise UdsError('x %r / e %r do not match the network (N=%d, E=%d)' % (tuple(x.shape), tuple(e.shape),
                                                                            net.graph.n_node, net.graph.n_edge))
    if S == 0:
        _dev(x, 'x')
        return (torch.empty((0, N, d), device=x.device, dtype=torch.float32),
                torch.empty((0, E, d), device=x.device, dtype=torch.float32))
    sp = _spatial_params(p)
    ws = torch.empty(lib.uds_spatial_workspace_floats(net.ptr, S, h, d), device=x.device, dtype=torch.float32)
    out_x = torch.empty((S, N, d), device=x.device, dtype=torch.float32)
    out_e = torch.empty((S, E, d), device=x.device, dtype=torch.float32)
    if rem_x is not None or rem_e is not None:
        if xb is not None or eb is not None or rem_x is None or rem_e is None:
            raise UdsError('rem_x and rem_e come together, without xb / eb')
        if tuple(rem_x.shape) != (S, N, h) or tuple(rem_e.shape) != (S, E, h):
            raise UdsError('rem_x %r / rem_e %r must be %r / %r' % (tuple(rem_x.shape), tuple(rem_e.shape), (S, N, h), (S, E, h)))
        _check(lib.uds_spatial_layer_forward_rem(net.ptr, ctypes.byref(sp), _dev(x, 'x'), fx, _dev(e, 'e'), fe, _dev(rem_x, 'rem_x'),
                                                 _dev(rem_e, 'rem_e'), S, h, d, ACT[act], int(flags), _dev(ws, 'workspace'),
                                                 _dev(out_x, 'out_x'), _dev(out_e, 'out_e'), _stream()), 'uds_spatial_layer_forward_rem')
        return out_x, out_e
    if xb is not None or eb is not None:
        for t, ref, name in ((xb, x, 'xb'), (eb, e, 'eb')):
            if t is not None and tuple(t.shape[:2]) != tuple(ref.shape[:2]):
                raise UdsError('%s %r does not match %r' % (name, tuple(t.shape), tuple(ref.shape)))
        _check(lib.uds_spatial_layer_forward_split(net.ptr, ctypes.byref(sp), _dev(x, 'x'), fx, _dev(xb, 'xb', True),
                                                   0 if xb is None else xb.shape[-1], _dev(e, 'e'), fe, _dev(eb, 'eb', True),
                                                   0 if eb is None else eb.shape[-1], S, h, d, ACT[act], int(flags),
                                                   _dev(ws, 'workspace'), _dev(out_x, 'out_x'), _dev(out_e, 'out_e'), _stream()),
               'uds_spatial_layer_forward_split')
        return out_x, out_e
    _check(lib.uds_spatial_layer_forward(net.ptr, ctypes.byref(sp), _dev(x, 'x'), fx, _dev(e, 'e'), fe, S, h, d, ACT[act],
                                         int(flags), _dev(ws, 'workspace'), _dev(out_x, 'out_x'), _dev(out_e, 'out_e'), _stream()),
           'uds_spatial_layer_forward')
    return out_x, out_e
