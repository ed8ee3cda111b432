"""Batch-dimension sharding of one axis transform across the GPUs of a node.

Every lane is transformed independently (the reference's own `_par` path relies on exactly that:
Zip::par_for_each over lanes, src/lib.rs:190-194), so a single-axis nd* call shards with NO
exchange step: split the lane index space into contiguous blocks along a non-transform dimension,
one block per rank (one process per GPU), run the same kernels on each shard.

RCCL (torch.distributed backend "nccl") is used only to move batch slices between a root and the
ranks -- scatter before / gather after -- over xGMI; `scatter_lanes` / `gather_lanes` are plain
point-to-point groups (one contiguous slice per peer link), never a ring.  Arrays that are born
sharded (each rank produces / consumes its own block) need no communication at all: call the nd*
function on the local shard.
"""
import numpy as np

try:
    import torch
    import torch.distributed as dist
except Exception:  # pragma: no cover
    torch = None
    dist = None


def shard_dim(shape, axis):
    """Dimension to split: the outermost non-transform dimension with extent > 1 (slices of a
    C-layout array along it are contiguous byte ranges)."""
    for d, e in enumerate(shape):
        if d != axis and e > 1:
            return d
    raise ValueError("array has a single lane: nothing to shard")


def shard_bounds(extent, world):
    """Contiguous blocks [lo, hi) per rank; the first `extent % world` ranks get one extra row."""
    base, rem = divmod(extent, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def local_shape(shape, axis, rank, world):
    d = shard_dim(shape, axis)
    lo, hi = shard_bounds(shape[d], world)[rank]
    s = list(shape)
    s[d] = hi - lo
    return tuple(s), d, (lo, hi)


def _slice(t, d, lo, hi):
    idx = [slice(None)] * t.dim()
    idx[d] = slice(lo, hi)
    return t[tuple(idx)]


def _exchange(sends, recvs, group=None):
    """One group of pairwise isend / irecv: `sends` = [(tensor, peer)], `recvs` = [(tensor, peer)] (contiguous tensors).
    RCCL ("nccl") moves device tensors over xGMI directly.  The gloo backend only carries host memory: device tensors are
    staged through host copies there (what the 2-rank tests on ONE GPU use -- RCCL refuses two ranks on one device --, so
    that the exchange logic runs with device-resident shards and the HIP kernels as the local executor)."""
    if not sends and not recvs:
        return
    stage = dist.get_backend(group) == "gloo"
    ops, host_recv, keep = [], [], []
    for t, peer in sends:
        w = t.cpu() if (stage and t.is_cuda) else t
        keep.append(w)
        ops.append(dist.P2POp(dist.isend, w, peer, group))
    for t, peer in recvs:
        if stage and t.is_cuda:
            h = torch.empty(t.shape, dtype=t.dtype, device="cpu")
            host_recv.append((t, h))
            ops.append(dist.P2POp(dist.irecv, h, peer, group))
        else:
            ops.append(dist.P2POp(dist.irecv, t, peer, group))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    for t, h in host_recv:
        t.copy_(h)


def scatter_lanes(full, shape, dtype, axis, root=0, device=None, group=None):
    """Root holds `full` (torch tensor of `shape`); every rank returns its contiguous shard.
    Point-to-point: root posts one isend per peer, peers one irecv (7 concurrent xGMI links from
    the root on an 8-GPU node)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lshape, d, _ = local_shape(shape, axis, rank, world)
    bounds = shard_bounds(shape[d], world)
    if rank == root:
        sends = []
        mine = None
        for r, (lo, hi) in enumerate(bounds):
            piece = _slice(full, d, lo, hi).contiguous()
            if r == root:
                mine = piece.clone() if piece.data_ptr() == full.data_ptr() else piece
            elif piece.numel():
                sends.append((piece, r))
        _exchange(sends, [], group)
        return mine
    out = torch.empty(lshape, dtype=dtype, device=device)
    if out.numel():
        _exchange([], [(out, root)], group)
    return out


def gather_lanes(local, full_shape, axis_out_shape_dim, root=0, out=None, group=None):
    """Inverse of scatter_lanes for the OUTPUT array: `full_shape` is the output's global shape and
    `axis_out_shape_dim` the dimension it was sharded along.  Root returns the assembled tensor."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    d = axis_out_shape_dim
    bounds = shard_bounds(full_shape[d], world)
    if rank == root:
        if out is None:
            out = torch.empty(full_shape, dtype=local.dtype, device=local.device)
        recvs, bufs = [], []
        for r, (lo, hi) in enumerate(bounds):
            if r == root:
                _slice(out, d, lo, hi).copy_(local)
                continue
            s = list(full_shape); s[d] = hi - lo
            buf = torch.empty(s, dtype=local.dtype, device=local.device)
            if buf.numel():
                bufs.append((buf, lo, hi))
                recvs.append((buf, r))
        _exchange([], recvs, group)
        for buf, lo, hi in bufs:
            _slice(out, d, lo, hi).copy_(buf)
        return out
    if local.numel():
        _exchange([(local.contiguous(), root)], [], group)
    return None


def transform_sharded(fn, full_in, in_shape, in_dtype, out_shape, out_dtype, handler, axis, root=0, device=None,
                      group=None):
    """root-scatter -> local nd* on every rank -> root-gather.  `fn` is one of the nd* functions
    (or, in the CPU/gloo tests, a stand-in with the same signature).  Returns the full output on
    root, None elsewhere.  NOTE: for a single transform the two transfers cost more than
    the transform; the scalable use is arrays that stay sharded across many nd* calls."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    x = scatter_lanes(full_in if rank == root else None, in_shape, in_dtype, axis, root, device, group)
    lshape_out, d, _ = local_shape(out_shape, axis, rank, world)
    y = torch.zeros(lshape_out, dtype=out_dtype, device=device)
    if x.numel() and y.numel():
        fn(x, y, handler, axis)
    return gather_lanes(y, tuple(out_shape), d, root, None, group)


# ---- multi-axis transforms on a sharded array ----------------------------------------
#
# A 2-D / n-D transform (the reference's examples/fft2.rs, rfft2.rs: axis 1, then axis 0 through a
# `work` array) on an array that is sharded along dimension `d` needs ONE exchange when the next
# transform axis IS `d`: re-shard from slabs along `d` to slabs along another dimension `e`.  That is
# an all-to-all: rank r sends the block (its rows of `d`) x (rank q's range of `e`) to every q.  On
# MI355X xGMI is a full point-to-point mesh (7 links per GPU), so the exchange is issued as ONE group
# of pairwise isend/irecv (RCCL groups them into a single launch; every link carries exactly one
# block in each direction) -- no ring, no staging through a root.  Uneven extents are allowed.

def reshard(local, global_shape, d_from, d_to, group=None):
    """`local` is this rank's slab of a `global_shape` array sharded along `d_from` (bounds from
    shard_bounds).  Returns this rank's slab of the SAME array sharded along `d_to`."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if d_from == d_to:
        return local
    bf = shard_bounds(global_shape[d_from], world)
    bt = shard_bounds(global_shape[d_to], world)
    oshape = list(global_shape); oshape[d_to] = bt[rank][1] - bt[rank][0]
    out = torch.empty(oshape, dtype=local.dtype, device=local.device)
    sends, recvs, placed = [], [], []
    for q in range(world):
        # block that stays on / goes to q: my d_from rows, q's d_to range
        piece = _slice(local, d_to, bt[q][0], bt[q][1])
        if q == rank:
            _slice(out, d_from, bf[rank][0], bf[rank][1]).copy_(piece)
            continue
        if piece.numel():
            sends.append((piece.contiguous(), q))
        rshape = list(oshape); rshape[d_from] = bf[q][1] - bf[q][0]
        buf = torch.empty(rshape, dtype=local.dtype, device=local.device)
        if buf.numel():
            placed.append((buf, bf[q][0], bf[q][1]))
            recvs.append((buf, q))
    _exchange(sends, recvs, group)
    for buf, lo, hi in placed:
        _slice(out, d_from, lo, hi).copy_(buf)
    return out


def transform_axes_sharded(steps, local, global_shape, sharded_dim, group=None, alloc=None):
    """Apply a sequence of single-axis transforms to an array that lives sharded across the ranks.

    steps         list of (fn, handler, axis, out_extent, out_dtype): fn is an nd* function,
                  out_extent the output length along `axis` (n, or n/2+1 for R2C, ...)
    local         this rank's slab (sharded along `sharded_dim`)
    Returns (local_out, out_global_shape, out_sharded_dim).  Whenever the next axis is the sharded
    dimension the array is first re-sharded (one all-to-all, `reshard`) to the outermost other
    dimension; otherwise the step runs on the local slab with no communication.  The result is left
    in whatever sharding the last step used (callers that need the original sharding call
    `reshard` once more) -- the same choice pencil/slab FFT codes make to save an exchange."""
    shape = list(global_shape)
    d = sharded_dim
    x = local
    for fn, handler, axis, out_extent, out_dtype in steps:
        if axis == d:
            e = next((k for k, ext in enumerate(shape) if k != axis and ext > 1), None)
            if e is None:
                raise ValueError("array has a single lane: nothing to re-shard to")
            x = reshard(x, tuple(shape), d, e, group)
            d = e
        oshape = list(x.shape); oshape[axis] = out_extent
        y = (alloc or torch.zeros)(oshape, dtype=out_dtype, device=x.device)
        if x.numel() and y.numel():
            fn(x.contiguous(), y, handler, axis)
        shape[axis] = out_extent
        x = y
    return x, tuple(shape), d
