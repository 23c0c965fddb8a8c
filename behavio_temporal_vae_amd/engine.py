"""Host-side sequencing of the HIP kernels for one conditional-VAE step.

The engine owns no parameters: it walks the module tree of a distribution network (same layout as
the reference's ``ParameterizerNN``, src/submodels.py:41-69), and for every Linear(+ReLU+BatchNorm)
unit issues the matching C-ABI calls on the current HIP stream.  torch only provides device buffers.

Forward of one unit in train mode:  GEMM(bias+ReLU, column-statistics epilogue) -> statistics
finalize -> BatchNorm apply (+ residual), or no apply at all when every reader is a plain Linear
(the reader folds the normalisation into its weights).  Eval mode folds BatchNorm into the GEMM epilogue.
Backward: BatchNorm/ReLU backward -> wgrad GEMM (split reduction) -> dgrad GEMM whose epilogue adds
the residual / second-consumer gradient and, where it can, runs the producing unit's BatchNorm
backward or its reduction pass.  The environment switches below turn the fusions off for A/B runs.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

from . import _native as nv

P = nv.ptr
PR = nv.ptr_rows
BN_EPS_DEFAULT = 1e-5


@dataclass
class Unit:
    """Linear (+ReLU+BatchNorm[+Dropout]) unit of a network."""
    lin: torch.nn.Linear
    bn: Optional[torch.nn.BatchNorm1d]
    act: bool            # ReLU (and BN when present) follow the Linear
    residual: bool
    dropout_p: float = 0.0


class BwdProducts:
    """Column reductions a BatchNorm's consumers leave while computing their weight gradients
    (``btvae_linear_wgrad_affine``): ``wp[n] = sum_m G*Y``, ``wb[n] = sum_m G`` for the gradient G that reaches the
    BatchNorm output.  Once every consumer has added its share, the BatchNorm backward needs no reduction pass."""

    def __init__(self, n: int, device):
        self.wp = torch.empty(n, device=device)
        self.wb = torch.empty(n, device=device)
        self.consumers = 0          # Linears that read the Normed in the forward
        self.filled = 0             # of which have run their weight gradient
        self.owner = None           # (Unit, Saved) of the BatchNorm unit
        self.dpre = None            # set once the last consumer's dX product has run that unit's BatchNorm backward
        self.db_cur = None


@dataclass
class Normed:
    """Output of a train-mode BatchNorm kept as its input plus the normalisation: ``y * scale + shift`` column-wise,
    never written out.  Linear consumers fold scale/shift into their weights (``btvae_bn_fold_linear``)."""
    y: torch.Tensor
    mean: torch.Tensor
    invstd: torch.Tensor
    gamma: torch.Tensor
    beta: torch.Tensor
    products: Optional["BwdProducts"] = None     # filled by the consumers' weight gradients in the backward

    @property
    def shape(self):
        return self.y.shape

    @property
    def device(self):
        return self.y.device


@dataclass
class Saved:
    x_in: torch.Tensor                       # input of the unit's Linear, or the tensor a Normed input is built on
    y: Optional[torch.Tensor] = None
    mean: Optional[torch.Tensor] = None
    invstd: Optional[torch.Tensor] = None
    keep: Optional[torch.Tensor] = None      # uint8 dropout keep flags of the unit's output
    keep_scale: float = 1.0
    total_rows: int = 0                      # rows the BatchNorm statistics span (> local rows under cross-rank sync)
    x_scale: Optional[torch.Tensor] = None   # Normed input: its column scale / shift (x = x_in * x_scale + x_shift)
    x_shift: Optional[torch.Tensor] = None
    x_products: Optional[BwdProducts] = None # Normed input: where this unit's weight gradient leaves its by-products
    out_products: Optional[BwdProducts] = None   # this unit returned a Normed: by-products its consumers fill
    x_src: Optional["Saved"] = None          # record of the BatchNorm unit that wrote this unit's (materialised) input
    readers: int = 0                         # Linears that read this unit's materialised output in the forward
    readers_done: int = 0                    # of which have run their dX product in the backward
    g_partials: Optional[tuple] = None       # (partials, row tiles) of sum G, sum G*xhat left by the last reader's dX product
    w_planes: Optional[torch.Tensor] = None  # bf16 planes of the unit's (unfolded) weight for the dX product, split mode
    synced: bool = False                     # the statistics were merged over a process group: the backward sums are too


class NetPlan:
    """Ordered units of one distribution network and its output non-linearities."""

    def __init__(self, dist):
        blocks = dist.parameterizer.block_dict
        self.dist = dist
        self.trunk = [self._unit(blocks["input"].input_layer, False)]
        self.trunk += [self._unit(seq, True) for seq in blocks["input"].middle_layers]
        self.trunk.append(self._unit(blocks["input"].output_layer, False))
        self.heads: Dict[str, List[Unit]] = {}
        for name in dist.parameterizer.dist_params:
            blk = blocks[name]
            units = [self._unit(blk.input_layer, False)]
            units += [self._unit(seq, True) for seq in blk.middle_layers]
            units.append(self._unit(blk.output_layer, False))
            self.heads[name] = units
        self.in_dim = self.trunk[0].lin.in_features

    @staticmethod
    def _unit(seq, residual) -> Unit:
        lin = seq[0]
        bn, act, p = None, False, 0.0
        for m in list(seq)[1:]:
            if isinstance(m, torch.nn.ReLU):
                act = True
            elif isinstance(m, torch.nn.BatchNorm1d):
                bn = m
            elif isinstance(m, torch.nn.Dropout):
                p = m.p
        return Unit(lin, bn, act, residual, p)

    def bn_modules(self):
        for u in self.trunk + [u for h in self.heads.values() for u in h]:
            if u.bn is not None:
                yield u.bn


class Workspace:
    """Reusable device scratch (split-reduction slabs, BatchNorm partials, norm partials)."""

    def __init__(self):
        self.bufs: Dict[tuple, torch.Tensor] = {}

    def get(self, key: str, numel: int, device, dtype=torch.float32) -> torch.Tensor:
        k = (key, str(device), dtype)
        t = self.bufs.get(k)
        if t is None or t.numel() < numel:
            t = torch.empty(max(numel, 1), device=device, dtype=dtype)
            self.bufs[k] = t
        return t


WS = Workspace()
WGRAD_WS_FLOATS = 32 * 1024 * 1024
HEAD_WS_FLOATS = 8 * 1024 * 1024


class DropoutFeed:
    """Source of train-mode dropout keep flags (torch.nn.Dropout at src/submodels.py:21): Philox draws on the
    device seeded from torch's CPU generator, or -- for parity tests -- flags queued in ``injected`` and consumed
    in the order the units run."""

    def __init__(self):
        self.injected: Optional[List[torch.Tensor]] = None

    def keep(self, M: int, N: int, p: float, dev) -> torch.Tensor:
        if self.injected is not None:
            if not self.injected:
                raise RuntimeError("dropout: injected keep-flag queue is empty")
            k = self.injected.pop(0)
            if tuple(k.shape) != (M, N) or k.dtype != torch.uint8 or k.device != dev or not k.is_contiguous():
                raise ValueError(f"dropout: injected keep flags must be contiguous uint8 [{M},{N}] on {dev}")
            return k
        k = torch.empty(M, N, dtype=torch.uint8, device=dev)
        seed = int(torch.empty((), dtype=torch.int64).random_().item())
        nv.call("btvae_dropout_keep_mask", P(k), M * N, float(p), seed, 0)
        return k


DROPOUT = DropoutFeed()
BN_SYNC = None      # parallel.BatchNormSync while BatchNorm statistics span the ranks of a process group, else None


def _check_train_supported(u: Unit, train: bool):
    if train and u.dropout_p >= 1.0:
        raise ValueError("dropout probability must be below 1")


FOLD_BN_INTO_CONSUMER = True     # train mode: leave BatchNorm outputs that only feed plain Linears unmaterialised
BWD_SUMS_FROM_CONSUMERS = True   # ... and take their backward column sums from those Linears' weight gradients
BWD_SUMS_IN_DGRAD = not os.environ.get('BTVAE_NO_DGRAD_SUMS')   # reduction pass of the other units' BatchNorm backward in the dX product that forms G
BWD_FUSED_INTO_CONSUMER = not os.environ.get('BTVAE_NO_FUSED_BWD')   # ... and run their BatchNorm backward in the epilogue of the last such Linear's dX product


PITCHED_ROWS = not os.environ.get('BTVAE_NO_PITCHED_ROWS')    # hidden activations and gradients in line-aligned rows (_rows_on_lines)
PRESPLIT_WEIGHTS = not os.environ.get('BTVAE_NO_PRESPLIT')   # split mode: weights split into bf16 planes once per pass


HIDDEN_BF16 = not os.environ.get('BTVAE_NO_BF16_STORAGE')   # bf16 mode: hidden activations and gradients are stored as bf16


PLANES_FWD, PLANES_DX = 1, 2


def _hidden_bf16() -> bool:
    """bf16 mode: the [rows, hidden width] activations and gradients between the Linear layers live in HBM as bf16 --
    the values the bf16 products multiply anyway -- so every kernel that touches them moves half the bytes, and the
    hidden-layer products copy them global -> LDS without converting.  Network inputs, head outputs, weights, statistics
    and every reduction stay fp32."""
    return HIDDEN_BF16 and nv.lib().btvae_get_gemm_precision() == 1


def _is16(t) -> bool:
    return t is not None and t.dtype == torch.bfloat16


def _bf16(a=None, out=None, aux=(), b=None):
    """Tell the next launch which of its matrices are stored as bf16 (``btvae_next_bf16``): the A operand, the output,
    the add-in / BatchNorm input (which must agree), a weight gradient's B operand."""
    aux = [t for t in aux if t is not None]
    if aux and len({t.dtype for t in aux}) != 1:
        raise RuntimeError("the add-in and the BatchNorm input of one launch disagree on their storage type")
    m = (1 if _is16(a) else 0) | (2 if _is16(out) else 0) | (4 if aux and _is16(aux[0]) else 0) | (8 if _is16(b) else 0)
    if m:
        nv.lib().btvae_next_bf16(m)


def _split_mode() -> int:
    """0 outside the split modes, else the number of pieces per operand value a Linear product may use (3 or 2)."""
    return {2: 3, 3: 2}.get(nv.lib().btvae_get_gemm_precision(), 0)


class _MaxSlots:
    """Device scalars for the absolute maxima of GEMM operand tensors (two-piece mode), handed out from zero-filled
    chunks; a slot lives as long as something references its view."""

    CHUNK = 256          # floats: 128 slots of two (max |t|, range statistic of the two-piece guard)

    def __init__(self):
        self.chunk, self.used = None, 0
        self.held, self.held_prev = [], []

    def fresh(self):
        """Start a new chunk with the next slot.  Every pass does: a pass captured into a HIP graph must zero its slots
        inside the graph, or a replay would extend the previous replay's maxima instead of taking its own.
        The chunks of the pass that ends here stay referenced until the end of the next one: a product is handed raw
        pointers into them (``_operand_maxima``), possibly through views that die before its launch is enqueued, and
        memory the allocator has taken back could be zero-filled for someone else in between."""
        self.held_prev = self.held + ([self.chunk] if self.chunk is not None else [])
        self.held = []
        self.chunk = None

    def take(self, dev) -> torch.Tensor:
        if self.chunk is None or self.used == self.CHUNK or self.chunk.device != dev:
            if self.chunk is not None:
                self.held.append(self.chunk)
            self.chunk, self.used = torch.zeros(self.CHUNK, device=dev), 0      # a fill kernel (a captured memset node has no
                                                                                 # effect on memory of the graph's own allocation node)
        self.used += 2
        return self.chunk[self.used - 2:self.used]


MAX_SLOTS = _MaxSlots()


def _amax(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """Device scalar holding max |t| (two-piece mode; None otherwise): the slot its producer kernel filled when the
    engine made ``t`` itself (``_track_output``; such tensors are written once), else a pass over ``t`` now -- network
    inputs, head gradients and weights may change between two uses without this module seeing it, so nothing is cached
    for them."""
    if t is None or _split_mode() != 2:
        return None
    slot = getattr(t, "_btvae_amax", None)
    if slot is None:
        slot = MAX_SLOTS.take(t.device)
        nv.call("btvae_absmax", P(t), t.numel(), P(slot))
    return slot


def _track_output(t: torch.Tensor):
    """Two-piece mode: have the producer kernel launched next leave max |t| in a fresh slot (``btvae_output_maximum``),
    so that the products reading ``t`` need no pass of their own over it."""
    if _split_mode() != 2:
        return
    slot = MAX_SLOTS.take(t.device)
    nv.lib().btvae_output_maximum(P(slot))
    t._btvae_amax = slot


def measure_once(t: torch.Tensor) -> torch.Tensor:
    """Two-piece mode: take max |t| now and keep it with ``t`` for every product that reads it.  For tensors the caller
    has just made and will not change (a network's concatenated input, the head gradients of a backward pass)."""
    if _split_mode() == 2 and getattr(t, "_btvae_amax", None) is None:
        t._btvae_amax = _amax(t)
    return t


class _Guard:
    """Watch over the two-piece products.  Each product site (a Linear's forward, dX or weight-gradient product) has an
    entry in a host-visible flag array the kernels write (``btvae_guard_flags``):

    * *range*: an operand had a region more than 2^16 below its tensor's maximum (``btvae_guard_range_log2``) -- its
      small values would keep an absolute, not a relative, error.  From the moment the host sees the flag, the site runs the three-piece kernel
      (no maxima handed over); ``fallbacks`` counts such launches.  The flags are read without synchronising, so the
      launch that raised one still ran on two pieces; with ``strict`` (``BTVAE_GUARD_STRICT=1``, or set by tests) the
      test is made by a launch of its own and waited for before every product, at the price of a device sync each.
    * *non-finite*: the accumulators of a two-piece product overflowed -- an operand maximum smaller than the data (a
      stale slot) or data that is itself non-finite.  ``check()`` raises; it is called where the host synchronises
      anyway (loss read-outs, validation, the benchmark's end)."""

    def __init__(self):
        self.ids: Dict[object, int] = {}
        self._flags = None
        self.fallbacks = 0
        self.strict = os.environ.get("BTVAE_GUARD_STRICT", "0") == "1"
        self.enabled = os.environ.get("BTVAE_GUARD", "1") != "0"       # 0: products are launched without a site (A/B timing)

    def flags(self):
        if self._flags is None:
            import numpy as np
            n = nv.lib().btvae_guard_sites()
            ptr = nv.lib().btvae_guard_flags()
            if not ptr:
                raise RuntimeError("guard flags are not set up (btvae_init has not run on any device)")
            self._flags = np.ctypeslib.as_array(ptr, shape=(2 * n,))
        return self._flags

    def site(self, key) -> int:
        s = self.ids.get(key)
        if s is None:
            s = self.ids[key] = len(self.ids) % nv.lib().btvae_guard_sites()     # beyond the table sites share entries
        return s

    def tripped(self, key, a: torch.Tensor, b: torch.Tensor) -> bool:
        """Is the range flag of ``key``'s site up?  ``strict``: after testing the operands with maxima ``a`` / ``b`` now."""
        s = self.site(key)
        f = self.flags()
        if self.strict and not f[2 * s] and not torch.cuda.is_current_stream_capturing():
            nv.call("btvae_guard_check", P(a), P(b), s)
            torch.cuda.current_stream().synchronize()
        return bool(f[2 * s])

    def pairs_ok(self, key, a: Optional[torch.Tensor], b: Optional[torch.Tensor], watch: bool = True) -> bool:
        """May ``key``'s product run on a pair-format operand (which has no three-piece form)?  Not once the site's range flag
        is up.  The kernels that read pairs test only the operands they split themselves, so the maxima given here (``a`` /
        ``b``: those of the other operand, e.g. the weight of a forward product) are tested by a launch of their own:
        waited for when ``strict``, else (``watch``) seen by the next pass; ``watch=False`` when the product tests them itself."""
        if not self.enabled:
            return True
        if self.tripped(key, a, b):
            self.fallbacks += 1
            return False
        if watch and not self.strict and (a is not None or b is not None):
            nv.call("btvae_guard_check", P(a), P(b), self.site(key))
        return True

    def tripped_sites(self) -> int:
        return int((self.flags()[0::2] != 0).sum())

    def check(self):
        f = self.flags()
        bad = (f[1::2] != 0).nonzero()[0]
        if len(bad):
            names = [k for k, v in self.ids.items() if v in set(bad.tolist())][:4]
            f[1::2] = 0
            raise RuntimeError(f"non-finite accumulators in a two-piece product (sites {names}): an operand's recorded maximum was "
                               f"smaller than its data, or the data itself is not finite")

    def reset(self):
        self.flags()[:] = 0
        self.fallbacks = 0


GUARD = _Guard()


def _operand_maxima(a: Optional[torch.Tensor], b: Optional[torch.Tensor], key=None, b_pairs: bool = False):
    """Hand the next product its operands' maxima (both or nothing) and its guard site; a site whose range flag is up
    gets neither and runs on three exact pieces.  ``b_pairs``: the B operand is in the pair format and has no maximum of its
    own (a weight gradient on a ``Pairs`` activation): only A's is handed over, and the guard watches A alone."""
    if a is None or (b is None and not b_pairs):
        return
    L = nv.lib()
    if key is None or not GUARD.enabled:
        L.btvae_gemm_operand_maxima(P(a), P(b))
        return
    s = GUARD.site(key)
    f = GUARD.flags()
    if GUARD.strict and not f[2 * s] and not torch.cuda.is_current_stream_capturing():
        nv.call("btvae_guard_check", P(a), P(b if b is not None else a), s)
        torch.cuda.current_stream().synchronize()
    if f[2 * s] and not b_pairs:                  # (a pair-format operand has no three-piece form to fall back to)
        GUARD.fallbacks += 1
        return
    L.btvae_gemm_operand_maxima(P(a), P(b))
    L.btvae_gemm_guard_site(s)


def _planes_amax(planes: Optional[torch.Tensor], W: torch.Tensor) -> Optional[torch.Tensor]:
    """Maximum that goes with a weight operand: the one its fp16 planes were scaled with, else computed from ``W`` now."""
    if _split_mode() != 2:
        return None
    if planes is not None:
        return planes._btvae_amax
    return _amax(W)


def _weight_image_bf16(W: torch.Tensor, which: int) -> Optional[torch.Tensor]:
    """bf16 storage mode: the stage images of ``W`` [N, K] for the 256-wide products on bf16-stored activations
    (``btvae_weight_image_bf16``): the forward image when the layer reads a hidden activation (K is the hidden width), the
    dX image when it writes one... a product whose image is missing runs the 128-wide kernel on the fp32 weight."""
    N, K = W.shape
    L = nv.lib()
    img = torch.empty(L.btvae_weight_image_bf16_bytes(N, K, 0) + L.btvae_weight_image_bf16_bytes(N, K, 1), dtype=torch.uint8, device=W.device)
    nv.call("btvae_weight_image_bf16", P(W), K, N, K, which, P(img))
    return img


def _weight_planes(W: torch.Tensor, which: int) -> Optional[torch.Tensor]:
    """bf16 planes of a weight matrix for the ``*_pw`` products (``btvae_weight_planes``; ``which``: the forward image,
    the dX image or both), or None outside the split mode.  They are a copy of ``W`` as it is now: made per forward
    pass, used by that pass and its backward."""
    mode = _split_mode()
    if _hidden_bf16():
        return _weight_image_bf16(W, which)
    if not PRESPLIT_WEIGHTS or not mode:
        return None
    N, K = W.shape
    planes = torch.empty(nv.lib().btvae_weight_planes_elems(N, K), dtype=torch.int16, device=W.device)
    if mode == 2:
        planes._btvae_amax = _amax(W)
        nv.call("btvae_weight_planes_f16", P(W), K, N, K, P(planes), which, P(planes._btvae_amax))
    else:
        nv.call("btvae_weight_planes", P(W), K, N, K, P(planes), which)
    return planes


def _plan_planes(plan: "NetPlan", backward: bool) -> Dict[torch.nn.Linear, torch.Tensor]:
    """bf16 planes of every Linear weight of a network in one launch (``btvae_weight_planes_multi``) at the start of a
    pass: hidden Linears get the forward image (+ the dX image when a backward follows), head output Linears the dX
    image only (their forward kernel reads the fp32 weight).  Empty outside the split mode."""
    mode = _split_mode()
    bf = _hidden_bf16()
    if not bf and (not PRESPLIT_WEIGHTS or not mode):
        return {}
    items = [(u.lin, PLANES_FWD | (PLANES_DX if backward else 0)) for u in plan.trunk]
    for units in plan.heads.values():
        items += [(u.lin, PLANES_FWD | (PLANES_DX if backward else 0)) for u in units[:-1]]
        if backward:
            items.append((units[-1].lin, PLANES_DX))
    L = nv.lib()
    if bf:
        # stage images for the 256-wide products: only layers that read or write a hidden activation in whole 64-byte row
        # pieces need them; the others' products read the fp32 weight
        sizes = [L.btvae_weight_image_bf16_bytes(lin.out_features, lin.in_features, 0) + L.btvae_weight_image_bf16_bytes(lin.out_features, lin.in_features, 1)
                 for lin, _ in items]
        buf = torch.empty(sum(sizes), dtype=torch.uint8, device=items[0][0].weight.device)
        out, off = {}, 0
        for (lin, _), n in zip(items, sizes):
            out[lin] = buf[off:off + n]       # sizes are multiples of 16 KiB
            off += n
        cnt = len(items)
        PtrArr, IntArr = nv.c_void_p * cnt, nv.ctypes.c_int * cnt
        nv.call("btvae_weight_image_bf16_multi", PtrArr(*[lin.weight.data_ptr() for lin, _ in items]), IntArr(*[lin.out_features for lin, _ in items]),
                IntArr(*[lin.in_features for lin, _ in items]), PtrArr(*[out[lin].data_ptr() for lin, _ in items]), IntArr(*[w for _, w in items]), cnt)
        return out
    sizes = [L.btvae_weight_planes_elems(lin.out_features, lin.in_features) for lin, _ in items]
    buf = torch.empty(sum(sizes), dtype=torch.int16, device=items[0][0].weight.device)
    out, off = {}, 0
    for (lin, _), n in zip(items, sizes):
        out[lin] = buf[off:off + n]           # every size is a multiple of 6144 elements: slices stay 16-byte aligned
        off += n
    cnt = len(items)
    PtrArr, IntArr = nv.c_void_p * cnt, nv.ctypes.c_int * cnt
    if mode == 2:
        maxima = torch.empty(2 * cnt, device=buf.device)      # two-float slots, cleared by the call itself
        for i, (lin, _) in enumerate(items):
            out[lin]._btvae_amax = maxima[2 * i:2 * i + 2]
        nv.call("btvae_weight_planes_f16_multi",
                PtrArr(*[lin.weight.data_ptr() for lin, _ in items]), IntArr(*[lin.out_features for lin, _ in items]),
                IntArr(*[lin.in_features for lin, _ in items]), PtrArr(*[out[lin].data_ptr() for lin, _ in items]),
                IntArr(*[w for _, w in items]), PtrArr(*[maxima[2 * i:2 * i + 2].data_ptr() for i in range(cnt)]), cnt)
        return out
    nv.call("btvae_weight_planes_multi",
            PtrArr(*[lin.weight.data_ptr() for lin, _ in items]), IntArr(*[lin.out_features for lin, _ in items]),
            IntArr(*[lin.in_features for lin, _ in items]), PtrArr(*[out[lin].data_ptr() for lin, _ in items]),
            IntArr(*[w for _, w in items]), cnt)
    return out


ROW_LINE_FLOATS = 32       # a 128-byte line


def _rows_on_lines(M: int, N: int, dev, on: Optional[bool] = None) -> torch.Tensor:
    """An [M, N] activation or gradient buffer whose rows start on 128-byte lines (pitch = N rounded up to 32 floats; a
    view when that pads).  The products read 64 bytes of each of 128 rows per step: from rows of 4000 bytes those pieces
    straddle sectors and lines (6 % of a product on real activations, 11 % at the full clock).  Every kernel that touches
    such a matrix is told its pitch: a leading-dimension argument where the entry point has one, ``_pitches`` otherwise.
    The pitch is a function of the width, so all [M, N] matrices of one call share it."""
    pitch = -(-N // ROW_LINE_FLOATS) * ROW_LINE_FLOATS
    if on is None and N % 8 == 0 and _hidden_bf16():
        # bf16 storage: the pitch is a multiple of 32 elements whatever PITCHED_ROWS says (the hidden-layer products copy
        # whole 64-byte pieces of a row; the producers keep the pad columns at zero)
        t = torch.empty(M, pitch, device=dev, dtype=torch.bfloat16)
        return t if pitch == N else t[:, :N]
    if pitch == N or not (PITCHED_ROWS if on is None else on):
        return torch.empty(M, N, device=dev)
    return torch.empty(M, pitch, device=dev)[:, :N]


def _rows_like(ref: torch.Tensor) -> torch.Tensor:
    """An uninitialised matrix of ``ref``'s shape and row pitch."""
    M, N = ref.shape
    if ref.stride(0) == N:
        return torch.empty(M, N, device=ref.device, dtype=ref.dtype)
    return torch.empty(M, ref.stride(0), device=ref.device, dtype=ref.dtype)[:, :N]


def _pitches(n: Optional[torch.Tensor] = None, k: Optional[torch.Tensor] = None, same_n=(), same_k=()):
    """Tell the next launch the row pitch of its [M, N] matrices (that of ``n``) and of its [M, K] ones (that of ``k``)
    when either is padded (``btvae_next_row_pitches``).  ``same_n`` / ``same_k``: the call's other matrices of those
    widths, which must have the same pitch (None entries are skipped)."""
    ln = n.stride(0) if n is not None else 0
    lk = k.stride(0) if k is not None else 0
    for group, ld in ((same_n, ln), (same_k, lk)):
        for t in group:
            if t is not None and t.stride(0) != ld:
                raise RuntimeError(f"matrices of one launch disagree on their row pitch: {tuple(t.shape)} has {t.stride(0)}, expected {ld}")
    if (n is not None and ln != n.shape[1]) or (k is not None and lk != k.shape[1]):
        nv.lib().btvae_next_row_pitches(ln, lk)


def _like_rows(t: Optional[torch.Tensor], ref: torch.Tensor) -> Optional[torch.Tensor]:
    """``t`` with the row pitch of ``ref`` (same shape): itself when it has it, else a copy (an input of the network that
    a residual unit adds to a padded activation)."""
    if t is None or (t.stride(0) == ref.stride(0) and t.dtype == ref.dtype):
        return t
    out = _rows_like(ref)
    if out.stride(0) != out.shape[1] and _is16(out):
        out.as_strided((out.shape[0], out.stride(0)), (out.stride(0), 1)).zero_()     # pad columns of a bf16 row stay zero
    out.copy_(t)
    if getattr(t, "_btvae_amax", None) is not None:
        out._btvae_amax = t._btvae_amax
    return out


EVAL_PAIRS = os.environ.get("BTVAE_EVAL_PAIRS", "1") != "0"
# rows from which the eval-mode chains keep their activations in the pair format: the 256-wide kernel that reads it wants
# several rounds of 256 x 256 tiles (at 25,000 rows x 1000 columns its 392 tiles are 1.5 rounds and the 128-wide kernel wins)
EVAL_PAIRS_MIN_ROWS = int(os.environ.get("BTVAE_EVAL_PAIRS_MIN_ROWS", "65536"))


# Training: the output of a BatchNorm-apply pass (a unit whose result has to exist in memory: it feeds a residual unit or is one)
# in the pair format when every reader is a BatchNorm unit's Linear -- that Linear's forward product and its weight gradient
# take the pieces as they are, a residual reader adds them.  From the same row count on as the eval-mode chains.
TRAIN_PAIRS = os.environ.get("BTVAE_TRAIN_PAIRS", "1") == "1"
TRAIN_PAIRS_MIN_ROWS = int(os.environ.get("BTVAE_TRAIN_PAIRS_MIN_ROWS", "65536"))


class Pairs:
    """A hidden activation in the pair format (csrc/pairs.h): ``data`` is an [M, N] fp32-typed view with a pitch that is a
    multiple of 32 floats whose bytes hold, per row and 16 columns, the 16 high and the 16 low fp16 pieces of the values;
    ``sc`` holds one power-of-two scale per 128 x 128 block.  The two-piece products take the pieces as they stand: eval-mode
    chains (scoring, imputation) keep the activations between two non-residual units in this form."""

    def __init__(self, data: torch.Tensor, sc: torch.Tensor):
        self.data, self.sc = data, sc

    @property
    def shape(self):
        return self.data.shape

    @property
    def device(self):
        return self.data.device

    def float(self) -> torch.Tensor:
        """The values as an fp32 matrix with the same row pitch (the pad columns decode to the zeros they hold), its maximum
        taken: what a consumer falls back to when its guard refuses the pair format; tests."""
        M, N = self.data.shape
        ld = self.data.stride(0)
        full = torch.empty(M, ld, device=self.data.device)
        nv.call("btvae_pairs_decode", PR(self.data), ld, P(self.sc), P(full), ld, M, ld if ld % 8 == 0 else (N + 7) // 8 * 8)
        out = full if ld == N else full[:, :N]
        if _split_mode() == 2:
            slot = MAX_SLOTS.take(full.device)
            nv.call("btvae_absmax", P(full), full.numel(), P(slot))
            out._btvae_amax = slot
        return out


_PAIR_CHAIN = {"refused": False}       # this pass over a network: the range guard has refused to start a pair-format chain


def _eval_pairs_ok(M: int, N: int) -> bool:
    """May an eval-mode unit with ``M`` rows of width ``N`` hand its output on in the pair format?  Two-piece mode, enough
    rows, rows on a pitch that is a multiple of 16 floats, 16-byte column groups."""
    pitch = -(-N // ROW_LINE_FLOATS) * ROW_LINE_FLOATS
    return (EVAL_PAIRS and M >= EVAL_PAIRS_MIN_ROWS and _split_mode() == 2 and PRESPLIT_WEIGHTS and N % 4 == 0
            and (PITCHED_ROWS or pitch == N))


def _train_pairs_ok(M: int, N: int) -> bool:
    """May a train-mode BatchNorm unit with ``M`` rows of width ``N`` write its materialised output in the pair format?
    The conditions of ``_eval_pairs_ok`` with the training switch and threshold."""
    pitch = -(-N // ROW_LINE_FLOATS) * ROW_LINE_FLOATS
    return (TRAIN_PAIRS and M >= TRAIN_PAIRS_MIN_ROWS and _split_mode() == 2 and PRESPLIT_WEIGHTS and N % 4 == 0
            and (PITCHED_ROWS or pitch == N))


def _weight_image_pq(W: torch.Tensor, wmax: torch.Tensor) -> torch.Tensor:
    """Stage images of ``W`` [N, K] for a product whose activation operand is in the pair format (``btvae_weight_image_pq``),
    scaled by the maximum in ``wmax``.  A copy of ``W`` as it is now: made per pass."""
    N, K = W.shape
    img = torch.empty(nv.lib().btvae_weight_image_pq_bytes(N, K, 0), dtype=torch.uint8, device=W.device)
    nv.call("btvae_weight_image_pq", P(W), K, N, K, 0, P(img), P(wmax))
    return img


def _linear_operands(lin: torch.nn.Linear, x):
    """(A, W, b, scale, shift) of ``lin`` applied to ``x``: a tensor as is, a Normed input through folded weights."""
    if not isinstance(x, Normed):
        return x, lin.weight, lin.bias, None, None
    if x.products is not None:
        x.products.consumers += 1
    N, K = lin.weight.shape
    dev = x.device
    Wf, bf = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
    sc, sh = torch.empty(K, device=dev), torch.empty(K, device=dev)
    _track_output(Wf)
    nv.call("btvae_bn_fold_linear", P(lin.weight), P(lin.bias), P(x.mean), P(x.invstd), P(x.gamma), P(x.beta), P(Wf), P(bf), P(sc), P(sh), N, K)
    return x.y, Wf, bf, sc, sh


def unit_forward(u: Unit, x, train: bool, saved: Optional[List[Saved]], keep_normed: bool = False,
                 planes: Optional[Dict[torch.nn.Linear, torch.Tensor]] = None, out_pairs: bool = False):
    """One Linear->ReLU->BatchNorm(->+x) unit (src/submodels.py:15-21,30-33) or a bare Linear.  ``x`` may be a
    ``Normed`` (train mode, non-residual units).  With ``keep_normed`` the caller promises that every consumer of
    the result is a plain (non-residual) Linear, and a BatchNorm unit without dropout returns a ``Normed``.
    ``planes``: pre-split weights of the network made at the start of the pass (``_plan_planes``).
    Eval mode: ``x`` may be a ``Pairs``; with ``out_pairs`` the caller promises that every consumer of the result is an
    eval-mode hidden unit, and the result is a ``Pairs`` when the unit can write one.
    Train mode: ``x`` may be a ``Pairs`` for a BatchNorm unit; with ``out_pairs`` the caller promises that every consumer is
    the Linear of a BatchNorm unit, and a materialised result (no ``Normed``, no dropout) is a ``Pairs`` from
    ``TRAIN_PAIRS_MIN_ROWS`` rows on.  A consumer whose guard site is flagged decodes its input and runs the fp32-operand path."""
    M, K = x.shape
    N = u.lin.out_features
    dev = x.device
    assert u.act, "bare Linear units are head outputs; use head_forward"
    _check_train_supported(u, train)
    if isinstance(x, Normed) and (u.residual or not train):
        raise RuntimeError("a residual unit needs its input materialised")
    x_pr = x.products if isinstance(x, Normed) else None
    x_src = getattr(x, "_btvae_src", None) if (train and saved is not None and not isinstance(x, Normed)) else None
    if x_src is not None:
        x_src.readers += 1
    x, W, b, x_sc, x_sh = _linear_operands(u.lin, x)
    # the dX product of the backward reads the unfolded weight
    same = W is u.lin.weight
    own = planes.get(u.lin) if planes else None            # planes of the unfolded weight, made with the rest of the network's
    if same:
        wp = own if own is not None else _weight_planes(W, PLANES_FWD | (PLANES_DX if saved is not None else 0))
        wp_bwd = wp if saved is not None else None
    else:
        wp = _weight_planes(W, PLANES_FWD)
        wp_bwd = None if saved is None else (own if own is not None else _weight_planes(u.lin.weight, PLANES_DX))
    if u.bn is not None and train:
        if M * (BN_SYNC.world if BN_SYNC is not None else 1) <= 1:
            # torch.nn.BatchNorm1d refuses a one-row batch in train mode (torch/nn/functional.py, _verify_batch_size)
            raise ValueError(f"Expected more than 1 value per channel when training, got input size torch.Size([{M}, {N}])")
        tiles = (M + 127) // 128
        y = _rows_on_lines(M, N, dev)
        part = WS.get("bnpart", tiles * 2 * N, dev)
        if isinstance(x, Pairs) and not GUARD.pairs_ok((id(u.lin.weight), "fwd"), None, _planes_amax(wp, W)):
            x = x.float()                             # the weight has regions far below its maximum: the guarded fp32-operand product
        if isinstance(x, Pairs):
            # the input was written in the pair format by the producing unit's BatchNorm-apply pass: the DMA-fed 256-wide kernel
            wmax = _planes_amax(wp, W)
            _track_output(y)
            nv.call("btvae_linear_relu_stats_fwd_pairs", PR(x.data), x.data.stride(0), P(x.sc), P(_weight_image_pq(W, wmax)), P(wmax), P(b),
                    PR(y), y.stride(0), P(part), M, K, N)
        else:
            _operand_maxima(_amax(x), _planes_amax(wp, W), (id(u.lin.weight), "fwd"))
            _track_output(y)
            _pitches(n=y)
            _bf16(a=x, out=y)
            nv.call("btvae_linear_relu_stats_fwd_pw", PR(x), x.stride(0), P(W), P(wp), P(b), PR(y), P(part), M, K, N)
        mean, invstd = torch.empty(N, device=dev), torch.empty(N, device=dev)
        track = u.bn.track_running_stats and u.bn.running_mean is not None
        mom = u.bn.momentum if u.bn.momentum is not None else 0.1
        total_rows = M
        if BN_SYNC is not None:
            local = torch.empty(2, N, device=dev)
            nv.call("btvae_bn_stats_local", P(part), tiles, M, N, P(local))
            parts = BN_SYNC.gather_stats(local)
            total_rows = M * BN_SYNC.world
            nv.call("btvae_bn_stats_merge", P(parts), BN_SYNC.world, M, N, P(mean), P(invstd),
                    P(u.bn.running_mean) if track else None, P(u.bn.running_var) if track else None, float(mom), float(u.bn.eps))
        else:
            nv.call("btvae_bn_stats_finalize", P(part), tiles, M, N, P(mean), P(invstd),
                    P(u.bn.running_mean) if track else None, P(u.bn.running_var) if track else None, float(mom), float(u.bn.eps))
        keep, ks = None, 1.0
        if u.dropout_p > 0.0:
            keep, ks = DROPOUT.keep(M, N, u.dropout_p, dev), 1.0 / (1.0 - u.dropout_p)
        normed = keep_normed and FOLD_BN_INTO_CONSUMER and keep is None and not u.residual
        out_pr = BwdProducts(N, dev) if (normed and saved is not None) else None
        if saved is not None:
            saved.append(Saved(x, y, mean, invstd, keep, ks, total_rows, x_sc, x_sh, x_pr, out_pr, x_src, w_planes=wp_bwd, synced=BN_SYNC is not None))
            if out_pr is not None:
                out_pr.owner = (u, saved[-1])
        if normed:
            return Normed(y, mean, invstd, u.bn.weight, u.bn.bias, out_pr)
        out = _rows_on_lines(M, N, dev)
        if out_pairs and keep is None and _train_pairs_ok(M, N) and not _is16(y):
            osc = torch.empty((M + 127) // 128, (N + 127) // 128, device=dev)
            r, r_ld, r_sc = None, 0, None
            if u.residual:
                r, r_ld, r_sc = (x.data, x.data.stride(0), x.sc) if isinstance(x, Pairs) else (x, x.stride(0), None)
            nv.call("btvae_bn_apply_fwd_pairs", PR(y), y.stride(0), P(mean), P(invstd), P(u.bn.weight), P(u.bn.bias), PR(r), r_ld, P(r_sc),
                    PR(out), out.stride(0), P(osc), M, N)
            res = Pairs(out, osc)
            if saved is not None:
                res._btvae_src = saved[-1]
            return res
        if isinstance(x, Pairs) and u.residual:
            x = x.float()                             # the streaming pass adds an fp32 matrix
        res_in = _like_rows(x, y) if u.residual else None
        _track_output(out)
        _pitches(n=y, same_n=(res_in, out))
        _bf16(out=out)
        nv.call("btvae_bn_apply_fwd", PR(y), P(mean), P(invstd), P(u.bn.weight), P(u.bn.bias), PR(res_in),
                P(keep), ks, PR(out), M, N)
        if saved is not None and keep is None:
            out._btvae_src = saved[-1]            # lets the Linear that reads `out` find this unit's record in the backward
    elif u.bn is None and train:
        # unit built without BatchNorm: y = relu(x W^T + b) is what the backward needs; dropout / residual on top
        if isinstance(x, Pairs):
            raise RuntimeError("a unit without BatchNorm reads an fp32 matrix (the caller promised BatchNorm readers)")
        y = _rows_on_lines(M, N, dev)
        _operand_maxima(_amax(x), _planes_amax(wp, W), (id(u.lin.weight), "fwd"))
        _track_output(y)
        _bf16(a=x, out=y)
        nv.call("btvae_linear_relu_affine_fwd_pw_ld", PR(x), x.stride(0), P(W), P(wp), P(b), None, None, None, 0, PR(y), y.stride(0), M, K, N)
        keep, ks = None, 1.0
        if u.dropout_p > 0.0:
            keep, ks = DROPOUT.keep(M, N, u.dropout_p, dev), 1.0 / (1.0 - u.dropout_p)
        if keep is not None or u.residual:
            out = _rows_on_lines(M, N, dev)
            res_in = _like_rows(x, y) if u.residual else None
            _track_output(out)
            _pitches(n=y, same_n=(res_in, out))
            _bf16(out=out)
            nv.call("btvae_bn_apply_fwd", PR(y), None, None, None, None, PR(res_in), P(keep), ks, PR(out), M, N)
        else:
            out = y
        if saved is not None:
            saved.append(Saved(x, y, None, None, keep, ks, 0, x_sc, x_sh, x_pr, None, x_src, w_planes=wp_bwd))
    else:
        out = _rows_on_lines(M, N, dev)
        scale = shift = None
        if u.bn is not None:
            scale, shift = torch.empty(N, device=dev), torch.empty(N, device=dev)
            nv.call("btvae_bn_fold_eval", P(u.bn.weight), P(u.bn.bias), P(u.bn.running_mean), P(u.bn.running_var), float(u.bn.eps), P(scale), P(shift), N)
        if isinstance(x, Pairs) and not GUARD.pairs_ok((id(u.lin.weight), "fwd"), None, _planes_amax(wp, W)):
            x = x.float()                             # as in train mode; the rest of the pass stays fp32
            _PAIR_CHAIN["refused"] = True
        pairs_in = isinstance(x, Pairs)
        out_pairs = out_pairs and (pairs_in or not u.residual) and _eval_pairs_ok(M, N)      # a residual unit adds a pair-format input only
        if out_pairs and not pairs_in and GUARD.enabled and (
                _PAIR_CHAIN["refused"] or GUARD.tripped((id(u.lin.weight), "fwd"), _amax(x), _planes_amax(wp, W))):
            # The chain would start here, and this product's range flag is up: its input has rows or columns far below the
            # tensor's maximum.  What such an input does to a hidden activation (one row 2^20 above its neighbours in a block)
            # a block scale cannot hold either, and a pair-format operand has no three-piece form: for the rest of this pass
            # over the network the activations stay fp32, where every product has its own guard.
            out_pairs = False
            _PAIR_CHAIN["refused"] = True
            GUARD.fallbacks += 1
        if pairs_in or out_pairs:
            if u.residual and not pairs_in:
                raise RuntimeError("a residual unit in the pair-format chain needs its input in that format")
            osc = torch.empty((M + 127) // 128, (N + 127) // 128, device=dev) if out_pairs else None
            wmax = _planes_amax(wp, W)
            if not out_pairs:
                _track_output(out)                    # an fp32 result is the operand of a head product: its maximum from this epilogue
            if pairs_in:
                nv.call("btvae_linear_relu_affine_fwd_pairs", PR(x.data), x.data.stride(0), P(x.sc), None, P(_weight_image_pq(W, wmax)), P(wmax),
                        P(b), P(scale), P(shift), 1 if u.residual else 0, PR(out), out.stride(0), P(osc), M, K, N)
            else:
                _operand_maxima(_amax(x), wmax, (id(u.lin.weight), "fwd"))
                nv.call("btvae_linear_relu_affine_fwd_pairs", PR(x), x.stride(0), None, P(W), P(wp), None,
                        P(b), P(scale), P(shift), 0, PR(out), out.stride(0), P(osc), M, K, N)
            return Pairs(out, osc) if out_pairs else out
        _operand_maxima(_amax(x), _planes_amax(wp, W), (id(u.lin.weight), "fwd"))
        _track_output(out)
        _bf16(a=x, out=out, aux=(x if u.residual else None,))
        nv.call("btvae_linear_relu_affine_fwd_pw_ld", PR(x), x.stride(0), P(W), P(wp), P(b), P(scale), P(shift),
                PR(x) if u.residual else None, x.stride(0), PR(out), out.stride(0), M, K, N)
    return out


def head_forward(u: Unit, x, act: int, lo: float, hi: float, want_dfac: bool, want_sq: bool,
                 planes: Optional[Dict[torch.nn.Linear, torch.Tensor]] = None):
    """Head output Linear fused with its clamp / softplus / relu non-linearity
    (src/submodels.py:102-104,156-157).  ``x`` may be a ``Normed``.
    Returns (value, dfac or None, value^2 or None, (scale, shift, by-product holder) of a Normed input or Nones,
    bf16 planes of the weight for the backward's dX product when ``want_dfac`` (split mode) or None)."""
    M, K = x.shape
    N = u.lin.out_features
    dev = x.device
    x_pr = x.products if isinstance(x, Normed) else None
    src = getattr(x, "_btvae_src", None) if not isinstance(x, Normed) else None
    if src is not None:
        src.readers += 1                      # its dX product does not leave the sums: the producer falls back to its reduction pass
    a, W, b, x_sc, x_sh = _linear_operands(u.lin, x)
    val = torch.empty(M, N, device=dev)
    dfac = torch.empty(M, N, device=dev) if want_dfac else None
    sq = torch.empty(M, N, device=dev) if want_sq else None
    ws = WS.get("headslab", HEAD_WS_FLOATS, dev) if (M <= 128 * 128 and not _is16(a)) else None      # few row tiles: split reduction (see the C side)
    if N > 64 and ws is None:                 # the 128-wide head kernel has a two-piece form; the narrow one (N <= 64) is
        _operand_maxima(_amax(a), _amax(W), (id(u.lin.weight), "fwd"))   # memory-bound and the split-reduction one latency-bound
    _bf16(a=a)
    nv.call("btvae_linear_head_fwd_ws", PR(a), a.stride(0), P(W), P(b), act, float(lo), float(hi), P(val), P(dfac), P(sq),
            P(ws), ws.numel() if ws is not None else 0, M, K, N)
    own = planes.get(u.lin) if planes else None
    return val, dfac, sq, (x_sc, x_sh, x_pr), ((own if own is not None else _weight_planes(u.lin.weight, PLANES_DX)) if want_dfac else None)


@dataclass
class NetTape:
    trunk: List[Saved] = field(default_factory=list)
    heads: Dict[str, List[Saved]] = field(default_factory=dict)
    head_in: Dict[str, torch.Tensor] = field(default_factory=dict)   # input of each head's output Linear (a Normed's y)
    head_in_affine: Dict[str, tuple] = field(default_factory=dict)    # (scale, shift) of that input when it was a Normed
    dfac: Dict[str, torch.Tensor] = field(default_factory=dict)
    head_planes: Dict[str, Optional[torch.Tensor]] = field(default_factory=dict)   # bf16 planes of each head's output weight


def net_forward(plan: NetPlan, x: torch.Tensor, train: bool, tape: Optional[NetTape], head_specs: Dict[str, tuple], want_sq=()):
    """Whole ParameterizerNN + output non-linearities.  head_specs[name] = (act, lo, hi).
    Returns {name: value}, {name: value^2} (for names in want_sq)."""
    # a unit's result may stay a Normed when everything that reads it is a plain Linear: the next unit of its chain
    # if that is not residual, the first unit of every head for the trunk's last unit, the head's output Linear
    h = x
    nt = len(plan.trunk)
    MAX_SLOTS.fresh()
    _PAIR_CHAIN["refused"] = False
    planes = _plan_planes(plan, tape is not None)
    for i, u in enumerate(plan.trunk):
        if i + 1 < nt:
            normed_ok = not plan.trunk[i + 1].residual
        else:
            normed_ok = all(not units[0].residual for units in plan.heads.values())
        # eval mode: the result may stay in the pair format when everything that reads it is a hidden unit (a residual one
        # adds it in that format)
        pairs_ok = not train and (i + 1 < nt or all(len(units) > 1 for units in plan.heads.values()))
        if train:
            # train mode: a materialised result may be in the pair format when every reader is the Linear of a BatchNorm unit
            readers = [plan.trunk[i + 1]] if i + 1 < nt else [units[0] for units in plan.heads.values() if len(units) > 1]
            pairs_ok = (i + 1 < nt or all(len(units) > 1 for units in plan.heads.values())) and all(r.bn is not None and r.act for r in readers)
        h = unit_forward(u, h, train, tape.trunk if tape is not None else None, keep_normed=train and normed_ok, planes=planes, out_pairs=pairs_ok)
    out, sqs = {}, {}
    for name, units in plan.heads.items():
        t = h
        sv = [] if tape is not None else None
        hidden = units[:-1]
        for i, u in enumerate(hidden):
            normed_ok = i + 1 == len(hidden) or not hidden[i + 1].residual
            t = unit_forward(u, t, train, sv, keep_normed=train and normed_ok, planes=planes,
                             out_pairs=not train and i + 1 < len(hidden))
        act, lo, hi = head_specs[name]
        val, dfac, sq, affine, hplanes = head_forward(units[-1], t, act, lo, hi, tape is not None, name in want_sq, planes)
        out[name] = val
        if sq is not None:
            sqs[name] = sq
        if tape is not None:
            tape.heads[name] = sv
            tape.head_in[name] = t.y if isinstance(t, Normed) else t
            tape.head_in_affine[name] = affine
            tape.dfac[name] = dfac
            tape.head_planes[name] = hplanes
    return out, sqs


def _grad_view(p: torch.nn.Parameter) -> torch.Tensor:
    if p.grad is None:
        raise RuntimeError("parameter has no gradient buffer (model arena not initialised)")
    return p.grad


def _wgrad(lin: torch.nn.Linear, dy: torch.Tensor, x: torch.Tensor, x_scale, x_shift, db_cur, accumulate: bool,
           products: Optional[BwdProducts] = None):
    """lin.weight.grad (+)= dy^T x', x' = x (plain input) or x * x_scale + x_shift (the input was a Normed; its
    producer's backward by-products are added to ``products``)."""
    M, N = dy.shape
    K = x.shape[1]
    ws = WS.get("wgrad", WGRAD_WS_FLOATS, dy.device)
    if isinstance(x, Pairs) and not GUARD.pairs_ok((id(lin.weight), "wgrad"), _amax(dy), None, watch=False):
        x = x.float()                                 # dY has tripped this site's range flag before: three exact pieces on fp32 operands
    if isinstance(x, Pairs):                          # the input was kept in the pair format: its pieces as they are
        if x_scale is not None:
            raise RuntimeError("a pair-format input is never a folded BatchNorm output")
        _operand_maxima(_amax(dy), None, (id(lin.weight), "wgrad"), b_pairs=True)
        _pitches(n=dy)
        nv.call("btvae_linear_wgrad_pairs", PR(dy), PR(x.data), x.data.stride(0), P(x.sc), P(_grad_view(lin.weight)), P(ws), ws.numel(), M, N, K, int(accumulate))
        return
    _operand_maxima(_amax(dy), _amax(x), (id(lin.weight), "wgrad"))
    _pitches(n=dy)
    _bf16(a=dy, b=x)
    if x_scale is None:
        nv.call("btvae_linear_wgrad", PR(dy), PR(x), x.stride(0), P(_grad_view(lin.weight)), P(ws), ws.numel(), M, N, K, int(accumulate))
    elif products is not None and BWD_SUMS_FROM_CONSUMERS:
        nv.call("btvae_linear_wgrad_affine", PR(dy), PR(x), x.stride(0), P(x_scale), P(x_shift), P(db_cur), P(lin.weight), P(products.wp), P(products.wb),
                int(products.filled > 0), P(_grad_view(lin.weight)), P(ws), ws.numel(), M, N, K, int(accumulate))
        products.filled += 1
    else:
        nv.call("btvae_linear_wgrad_affine", PR(dy), PR(x), x.stride(0), P(x_scale), P(x_shift), P(db_cur), None, None, None, 0,
                P(_grad_view(lin.weight)), P(ws), ws.numel(), M, N, K, int(accumulate))


def _fusable(pr: Optional[BwdProducts]) -> bool:
    """The producing BatchNorm's backward can run inside this (last) consumer's dX product: every consumer's weight
    gradient has left its by-products, and the fused kernel's alignment need holds."""
    if pr is None or not (BWD_SUMS_FROM_CONSUMERS and BWD_FUSED_INTO_CONSUMER) or pr.owner is None:
        return False
    _, sv_p = pr.owner
    return pr.consumers > 0 and pr.filled == pr.consumers and sv_p.keep is None and sv_p.y.shape[1] % 4 == 0


def _dgrad_with_producer_backward(pr: BwdProducts, dy: torch.Tensor, W: torch.Tensor, r: Optional[torch.Tensor], accumulate: bool,
                                  planes: Optional[torch.Tensor] = None):
    """dX product of a Linear that read a Normed, fused with the BatchNorm+ReLU backward of the unit that produced it.
    Returns that unit's dPre (what its own backward continues from); its BatchNorm and bias gradients are written."""
    u_p, sv_p = pr.owner
    M, N = dy.shape
    Np = sv_p.y.shape[1]
    dev = dy.device
    sums = torch.empty(2, Np, device=dev)
    nv.call("btvae_bn_bwd_sums_from_products", P(pr.wp), P(pr.wb), P(sv_p.mean), P(sv_p.invstd),
            P(_grad_view(u_p.bn.weight)), P(_grad_view(u_p.bn.bias)), P(sums), Np, int(accumulate))
    if sv_p.synced:
        BN_SYNC.sum_(sums)
    tiles = (M + 127) // 128
    dpre = _rows_like(sv_p.y)
    part = WS.get("bnbwd_part", tiles * Np, dev)
    r = _like_rows(r, dpre)
    _operand_maxima(_amax(dy), _planes_amax(planes, W), (id(W), "dgrad"))
    _track_output(dpre)
    _pitches(n=dy, k=dpre, same_k=(r, sv_p.y))
    _bf16(a=dy, out=dpre, aux=(r, sv_p.y))
    nv.call("btvae_linear_dgrad_bnbwd_pw", PR(dy), P(W), P(planes), PR(r), PR(sv_p.y), P(sv_p.mean), P(sv_p.invstd), P(u_p.bn.weight), P(sums),
            int(max(sv_p.total_rows, M)), PR(dpre), P(part), M, N, Np)
    db_cur = torch.empty(Np, device=dev) if sv_p.x_scale is not None else None
    nv.call("btvae_colsum_partials", P(part), tiles, Np, P(_grad_view(u_p.lin.bias)), P(db_cur), int(accumulate))
    pr.dpre, pr.db_cur = dpre, db_cur
    return dpre


def _unit_backward(u: Unit, sv: Saved, g: torch.Tensor, add_to: Optional[torch.Tensor], need_dx: bool, accumulate: bool):
    """Backward of a ReLU+BN unit.  g = dL/d(out).  Returns dL/d(x_in) (+ add_to, + g for residual units)."""
    M, N = g.shape
    K = sv.x_in.shape[1]
    dev = g.device
    pr = sv.out_products
    fused_done = pr is not None and pr.dpre is not None
    if fused_done:
        # the last consumer's dX product already ran this unit's BatchNorm backward: g is this unit's dPre
        if g is not pr.dpre:
            raise RuntimeError("fused BatchNorm backward: the gradient handed on is not the one the consumer produced")
        dpre, db_cur, scr = pr.dpre, pr.db_cur, None
    else:
        scr = WS.get("bnbwd", nv.lib().btvae_bn_relu_bwd_scratch(M, N), dev)
        dpre = _rows_like(sv.y)
        g = _like_rows(g, sv.y)
        db_cur = torch.empty(N, device=dev) if sv.x_scale is not None else None    # this pass's bias gradient, for the affine wgrad
    if fused_done:
        pass
    elif u.bn is None:
        _track_output(dpre)
        _pitches(n=sv.y, same_n=(g, dpre))
        _bf16(out=dpre)
        nv.call("btvae_relu_bwd", PR(g), PR(sv.y), P(sv.keep), float(sv.keep_scale), P(_grad_view(u.lin.bias)), P(db_cur), PR(dpre), P(scr), M, N, int(accumulate))
    elif pr is not None and pr.consumers > 0 and pr.filled == pr.consumers and sv.keep is None:
        # every reader of this unit's output was a folded Linear whose weight gradient has run: the column sums of
        # G and G*xhat follow from their by-products, no reduction pass over G and Y
        sums = torch.empty(2, N, device=dev)
        nv.call("btvae_bn_bwd_sums_from_products", P(pr.wp), P(pr.wb), P(sv.mean), P(sv.invstd),
                P(_grad_view(u.bn.weight)), P(_grad_view(u.bn.bias)), P(sums), N, int(accumulate))
        if sv.synced:
            BN_SYNC.sum_(sums)
        _track_output(dpre)
        _pitches(n=sv.y, same_n=(g, dpre))
        _bf16(out=dpre)
        nv.call("btvae_bn_relu_bwd_apply", PR(g), PR(sv.y), P(sv.mean), P(sv.invstd), P(u.bn.weight), None, 1.0,
                P(sums), int(max(sv.total_rows, M)), P(_grad_view(u.lin.bias)), P(db_cur), PR(dpre), P(scr), M, N, int(accumulate))
    elif sv.g_partials is not None:
        # the dX product that formed g left the per-tile column sums of g and g*xhat: no reduction pass over g and y
        part, tiles = sv.g_partials
        sums = torch.empty(2, N, device=dev)
        nv.call("btvae_bn_bwd_sums_from_partials", P(part), tiles, N, P(_grad_view(u.bn.weight)), P(_grad_view(u.bn.bias)), P(sums), int(accumulate))
        if sv.synced:
            BN_SYNC.sum_(sums)
        _track_output(dpre)
        _pitches(n=sv.y, same_n=(g, dpre))
        _bf16(out=dpre)
        nv.call("btvae_bn_relu_bwd_apply", PR(g), PR(sv.y), P(sv.mean), P(sv.invstd), P(u.bn.weight), None, 1.0,
                P(sums), int(max(sv.total_rows, M)), P(_grad_view(u.lin.bias)), P(db_cur), PR(dpre), P(scr), M, N, int(accumulate))
    elif sv.synced:
        # statistics span ranks: local column sums -> sum over ranks -> dPre with the batch-wide sums
        sums = torch.empty(2, N, device=dev)
        _pitches(n=sv.y, same_n=(g,))
        _bf16(a=g)
        nv.call("btvae_bn_relu_bwd_sums", PR(g), PR(sv.y), P(sv.mean), P(sv.invstd), P(sv.keep), float(sv.keep_scale),
                P(_grad_view(u.bn.weight)), P(_grad_view(u.bn.bias)), P(sums), P(scr), M, N, int(accumulate))
        BN_SYNC.sum_(sums)
        _track_output(dpre)
        _pitches(n=sv.y, same_n=(g, dpre))
        _bf16(out=dpre)
        nv.call("btvae_bn_relu_bwd_apply", PR(g), PR(sv.y), P(sv.mean), P(sv.invstd), P(u.bn.weight), P(sv.keep), float(sv.keep_scale),
                P(sums), int(sv.total_rows), P(_grad_view(u.lin.bias)), P(db_cur), PR(dpre), P(scr), M, N, int(accumulate))
    else:
        _track_output(dpre)
        _pitches(n=sv.y, same_n=(g, dpre))
        _bf16(out=dpre)
        nv.call("btvae_bn_relu_bwd", PR(g), PR(sv.y), P(sv.mean), P(sv.invstd), P(u.bn.weight), P(sv.keep), float(sv.keep_scale),
                P(_grad_view(u.bn.weight)), P(_grad_view(u.bn.bias)),
                P(_grad_view(u.lin.bias)), P(db_cur), PR(dpre), P(scr), M, N, int(accumulate))
    _wgrad(u.lin, dpre, sv.x_in, sv.x_scale, sv.x_shift, db_cur, accumulate, sv.x_products)
    if not need_dx:
        return None
    r = None
    if u.residual and add_to is not None:
        raise RuntimeError("residual unit with a second gradient consumer is not a layout this engine builds")
    if u.residual:
        r = g
    elif add_to is not None:
        r = add_to
    if _fusable(sv.x_products):
        return _dgrad_with_producer_backward(sv.x_products, dpre, u.lin.weight, r, accumulate, sv.w_planes)
    dx = _rows_like(sv.x_in.data if isinstance(sv.x_in, Pairs) else sv.x_in)     # a hidden activation's gradient has its pitch; the network input's is dense
    r = _like_rows(r, dx)
    src = sv.x_src
    if src is not None:
        src.readers_done += 1
    if (BWD_SUMS_IN_DGRAD and src is not None and src.readers_done == src.readers and src.mean is not None and src.keep is None
            and src.out_products is None and K % 4 == 0):
        tiles = (M + 127) // 128
        part = torch.empty(tiles, 2, K, device=dev)
        _operand_maxima(_amax(dpre), _planes_amax(sv.w_planes, u.lin.weight), (id(u.lin.weight), "dgrad"))
        _pitches(n=dpre, k=dx, same_k=(r, src.y))
        _bf16(a=dpre, out=dx, aux=(r, src.y))
        nv.call("btvae_linear_dgrad_gstats_pw", PR(dpre), P(u.lin.weight), P(sv.w_planes), PR(r), PR(src.y), P(src.mean), P(src.invstd), PR(dx), P(part), M, N, K)
        src.g_partials = (part, tiles)
    else:
        _operand_maxima(_amax(dpre), _planes_amax(sv.w_planes, u.lin.weight), (id(u.lin.weight), "dgrad"))
        _pitches(n=dpre)
        _bf16(a=dpre, out=dx, aux=(r,))
        nv.call("btvae_linear_dgrad_pw", PR(dpre), P(u.lin.weight), P(sv.w_planes), PR(r), PR(dx), dx.stride(0), M, N, K)
    return dx


def net_backward(plan: NetPlan, tape: NetTape, draw: Dict[str, torch.Tensor], need_input_grad: bool, accumulate: bool = False,
                 after_heads=None):
    """Backward through heads and trunk given dL/d(raw head outputs).  Parameter gradients are written
    (or accumulated) into the ``.grad`` views of the model's gradient arena.  ``after_heads()`` is called once every
    head's gradients are enqueued (data-parallel training starts reducing them while the trunk backward runs)."""
    g_trunk = None
    for name, units in plan.heads.items():
        d = measure_once(draw[name])              # read by this head's weight gradient and by its dX product
        M, N = d.shape
        dev = d.device
        last = units[-1]
        t_in = tape.head_in[name]
        K = t_in.shape[1]
        cs = WS.get("colsum", nv.lib().btvae_strip_chunks(M) * N, dev)
        x_sc, x_sh, x_pr = tape.head_in_affine.get(name, (None, None, None))
        db_cur = torch.empty(N, device=dev) if x_sc is not None else None
        nv.call("btvae_colsum", P(d), P(_grad_view(last.lin.bias)), P(db_cur), P(cs), M, N, int(accumulate))
        _wgrad(last.lin, d, t_in, x_sc, x_sh, db_cur, accumulate, x_pr)
        if _fusable(x_pr):
            g = _dgrad_with_producer_backward(x_pr, d, last.lin.weight, None, accumulate, tape.head_planes.get(name))
        else:
            g = _rows_like(t_in)
            _operand_maxima(_amax(d), _planes_amax(tape.head_planes.get(name), last.lin.weight), (id(last.lin.weight), "dgrad"))
            _bf16(a=d, out=g)
            nv.call("btvae_linear_dgrad_pw", P(d), P(last.lin.weight), P(tape.head_planes.get(name)), None, PR(g), g.stride(0), M, N, K)
        svs = tape.heads[name]
        for i in range(len(units) - 2, -1, -1):
            first = i == 0
            g = _unit_backward(units[i], svs[i], g, g_trunk if first else None, True, accumulate)
        g_trunk = g
    if after_heads is not None:
        after_heads()
    g = g_trunk
    for i in range(len(plan.trunk) - 1, -1, -1):
        g = _unit_backward(plan.trunk[i], tape.trunk[i], g, None, need_input_grad or i > 0, accumulate)
    return g


# -------------------------------------------------------------------------------------------------
# dictionary-Gaussian likelihood (covariance GEMM + per-row inversion kernel + gradient GEMMs)
# -------------------------------------------------------------------------------------------------
class DictTables:
    """Column-normalised dictionary and its packed outer-product table for one value of ``SigmaMapper``
    (src/submodels.py:151-152,176-179)."""

    def __init__(self, U: torch.Tensor, marginal_std_lim: float):
        D, V = U.shape
        dev = U.device
        self.D, self.V, self.NP = D, V, D * (D + 1) // 2
        self.Uhat = torch.empty(D, V, device=dev)
        self.norms = torch.empty(V, device=dev)
        self.UU = torch.empty(self.NP, V, device=dev)
        self.diag = torch.empty(self.NP, device=dev)
        nv.call("btvae_dict_prepare", P(U), P(self.Uhat), P(self.norms), P(self.UU), P(self.diag), D, V, float(marginal_std_lim))


def dict_covariance_packed(tab: DictTables, sigma_sq: torch.Tensor) -> torch.Tensor:
    rows = sigma_sq.shape[0]
    Sp = torch.empty(rows, tab.NP, device=sigma_sq.device)
    nv.call("btvae_matmul_nt_bias", P(sigma_sq), tab.V, P(tab.UU), tab.V, P(tab.diag), P(Sp), tab.NP, rows, tab.NP, tab.V)
    return Sp


def dict_loglik_fwd(tab: DictTables, x: torch.Tensor, mu: torch.Tensor, sigma_sq: torch.Tensor) -> torch.Tensor:
    """log N(x[row % B]; mu[row], Sigma[row]) for every row (src/submodels.py:169-171)."""
    rows, B = mu.shape[0], x.shape[0]
    Sp = dict_covariance_packed(tab, sigma_sq)
    logp = torch.empty(rows, device=mu.device)
    nv.call("btvae_mvn_rows_fwd", P(Sp), P(mu), P(x), P(logp), rows, B, tab.D)
    return logp


def dict_loglik_fwd_bwd(tab: DictTables, x, mu, sigma, sigma_sq, dfac_mu, dfac_sigma, g_const: float, dU: torch.Tensor, accumulate=False,
                        g_rows: Optional[torch.Tensor] = None):
    """Log-likelihood rows plus gradients wrt the raw decoder head outputs and the dictionary, for an
    upstream gradient ``g`` per row (constant, or ``g_rows``)."""
    rows, B = mu.shape[0], x.shape[0]
    dev = mu.device
    Sp = dict_covariance_packed(tab, sigma_sq)
    logp = torch.empty(rows, device=dev)
    dmu = torch.empty(rows, tab.D, device=dev)
    Pg = torch.empty(rows, tab.NP, device=dev)
    nv.call("btvae_mvn_rows_fwd_bwd", P(Sp), P(mu), P(x), P(dfac_mu), P(g_rows), float(g_const), P(logp), P(dmu), P(Pg), rows, B, tab.D)
    del Sp
    dsig = torch.empty(rows, tab.V, device=dev)
    nv.call("btvae_matmul_nn_mul", P(Pg), tab.NP, P(tab.UU), tab.V, P(sigma), P(dfac_sigma), tab.V, P(dsig), tab.V, rows, tab.V, tab.NP)
    ws = WS.get("wgrad", WGRAD_WS_FLOATS, dev)
    M1 = torch.empty(tab.NP, tab.V, device=dev)
    nv.call("btvae_matmul_tn", P(Pg), tab.NP, P(sigma_sq), tab.V, P(M1), P(ws), ws.numel(), tab.NP, tab.V, rows, 0, 3)
    nv.call("btvae_dict_grad", P(M1), P(tab.Uhat), P(tab.norms), P(dU), tab.D, tab.V, int(accumulate))
    return logp, dmu, dsig


def diag_loglik(x, mu, sigma, dfac_mu=None, dfac_sigma=None, g_const: float = 0.0, g_rows=None):
    rows, D = mu.shape
    B = x.shape[0]
    logp = torch.empty(rows, device=mu.device)
    if dfac_mu is None:
        nv.call("btvae_diag_gauss_loglik", P(x), P(mu), P(sigma), P(logp), None, None, None, None, None, 0.0, rows, B, D)
        return logp, None, None
    dm, ds = torch.empty_like(mu), torch.empty_like(sigma)
    nv.call("btvae_diag_gauss_loglik", P(x), P(mu), P(sigma), P(logp), P(dfac_mu), P(dfac_sigma), P(dm), P(ds), P(g_rows), float(g_const), rows, B, D)
    return logp, dm, ds


def reparam_kl(mu, sigma, eps, cond, S: int, three_d: bool, B: int, seed: int = 0, offset: int = 0):
    """z‖c rows for the decoder and per-row KL.  2-D conditions: posterior rows R=B, S draws each
    (src/vae_models.py:262-264).  3-D conditions ``cond[B,S,C]``: posterior rows R=S·B, one draw each
    (src/vae_models.py:269-274).  eps None -> in-kernel Philox draws (returned)."""
    R, L = mu.shape
    dev = mu.device
    C = cond.shape[-1] if cond is not None else 0
    Sl = 1 if three_d else S
    zc = torch.empty(Sl * R, L + C, device=dev)
    kl = torch.empty(R, device=dev)
    eps_out = None
    if eps is None:
        eps_out = torch.empty(Sl * R, L, device=dev)
    if three_d:
        cdiv, hi, lo = B, C, S * C
    else:
        cdiv, hi, lo = R, 0, C
    _track_output(zc)
    nv.call("btvae_reparam_kl_fwd", P(mu), P(sigma), P(eps), P(eps_out), P(cond), P(zc), P(kl), R, Sl, L, C, cdiv, hi, lo, int(seed), int(offset))
    return zc, kl, (eps if eps is not None else eps_out)


def reparam_kl_bwd(dzc, eps, mu, sigma, dfac_mu, dfac_sigma, S: int, three_d: bool, C: int, kl_weight: float):
    R, L = mu.shape
    Sl = 1 if three_d else S
    dm, ds = torch.empty_like(mu), torch.empty_like(sigma)
    nv.call("btvae_reparam_kl_bwd", P(dzc), P(eps), P(mu), P(sigma), P(dfac_mu), P(dfac_sigma), P(dm), P(ds), R, Sl, L, C, float(kl_weight))
    return dm, ds


def elbo_scalars(logp: torch.Tensor, klrow: torch.Tensor, beta: float) -> torch.Tensor:
    out = torch.empty(4, device=logp.device)
    nv.call("btvae_elbo_scalars", P(logp), logp.numel(), P(klrow), klrow.numel(), float(beta), P(out))
    return out


class ClipAdam:
    """State of the fused clip + Adam step over flat arenas (src/vae_models.py:103,323-324)."""

    def __init__(self, n: int, device):
        self.m = torch.zeros(n, device=device)
        self.v = torch.zeros(n, device=device)
        self.partials = torch.empty(512, dtype=torch.float64, device=device)
        self.norm = torch.zeros(1, device=device)
        self.step_count = 0

    def step(self, params: torch.Tensor, grads: torch.Tensor, n: int, lr: float, betas, eps: float, weight_decay: float, max_norm: Optional[float]):
        self.step_count += 1
        nv.call("btvae_grad_sumsq", P(grads), n, P(self.partials))
        nv.call("btvae_clip_adam_step", P(params), P(grads), P(self.m), P(self.v), n, P(self.partials),
                float(max_norm) if max_norm is not None else -1.0, float(lr), float(betas[0]), float(betas[1]), float(eps),
                float(weight_decay), self.step_count, P(self.norm))
