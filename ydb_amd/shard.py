"""Row-range sharding of a scan across the GPUs of one node and the merge of per-shard partial aggregates.

A column table is already split into shards that scan independently
(ydb/core/tx/columnshard/columnshard__scan.cpp:300-340 runs the program per batch and ships the
partial aggregates to the compute actor, which merges them).  With one process per GPU each rank runs the
program over its own rows; the only exchange is the small table of partial aggregates:

  * every rank serialises its result batch into one byte tensor,
  * one all-gather (RCCL over xGMI on GPUs, gloo on CPU) replicates the partial tables,
  * the partial tables are concatenated and reduced by a *merge program* -- an SSA GroupBy over the
    partial columns (SUM of sums and counts, MIN of mins, MAX of maxes, SOME of somes) with the same keys
    -- which runs through the same GPU executor as any other program.

High-cardinality groupings instead hash-partition the partial tables (`all_to_all_results`: point-to-point
sends to the rank that owns each key hash, `partition_of_keys`), so that each rank merges 1/N of the groups
and the result stays distributed.

On GPUs (`ShardedScan.run`) none of this touches the host: the partial table stays an HBM-resident batch, is laid
out as one flat device buffer (ssa_batch_pack_*), travels through one `all_gather_into_tensor` -- or, partitioned by
the XXH64 of its keys with the sharding kernels, one `all_to_all_single` -- over RCCL, and the received parts are
concatenated (one launch) into the input of the merge program, which runs through the same executor.  The only host
traffic is the parts' headers (a few hundred bytes per rank: row counts and buffer sizes).  The functions that pack
results on the host remain for ranks without a GPU (the CPU tests of this path).
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .arrow_c import Col, ResultColumn
from .ssa_pb import AGG, Aggregate, Assign, Column, Filter, Function, GroupBy, Program, Projection

Resolver = Sequence[Tuple[int, str]]

_PARTIAL_PREFIX = "partial:"
_MERGE_ID_BASE = 1_000_000


def _column_name(c, names: Dict[int, str]) -> str:
    """Name of a program column the way TProgramBuilder derives it (ydb/core/tx/program/program.cpp:60-90):
    the resolver's name for a table column, the explicit name, else the decimal id."""
    if isinstance(c, Column):
        if c.name is not None:
            return c.name
        c = c.id
    if isinstance(c, str):
        return c
    return names.get(int(c), str(int(c)))


def last_group_by(program: Program) -> Optional[GroupBy]:
    gb = None
    for cmd in program.commands:
        if isinstance(cmd, GroupBy):
            gb = cmd
    return gb


def split_at_last_group_by(program: Program) -> Tuple[Program, List[object]]:
    """(the program up to and including its last GroupBy, the commands after it).  Shards run the first part; what
    follows an aggregation (a HAVING filter, avg = sum / count, the final projection) only makes sense on merged
    groups and runs after the merge."""
    found = [i for i, cmd in enumerate(program.commands) if isinstance(cmd, GroupBy)]
    if not found:
        raise ValueError("program has no GroupBy: shards are concatenated, not merged")
    if len(found) > 1:
        # (the groups of the first aggregation span the shards: a second one over per-shard partial groups, merged
        # afterwards, is not the second aggregation of the whole table)
        raise ValueError("program has more than one GroupBy: only the last aggregation of a program can be merged across shards")
    idx = found[-1]
    return Program(commands=list(program.commands[:idx + 1]), version=program.version, kernels=program.kernels), list(program.commands[idx + 1:])


def merge_program(program: Program, resolver: Resolver) -> Tuple[Program, List[Tuple[int, str]]]:
    """Program + resolver that reduce concatenated partial results of `program` -- of its part up to the last GroupBy
    (split_at_last_group_by) -- to its final result: the merging GroupBy, then the program's own commands after the
    aggregation.

    Partial aggregate columns are fed in under the name "partial:<name>" so that the merged aggregate can
    carry the original name; key columns keep their names."""
    names = {i: n for i, n in resolver}
    gb = last_group_by(program)
    if gb is None:
        raise ValueError("program has no GroupBy: shards are concatenated, not merged")
    merge_of = {AGG.SUM: AGG.SUM, AGG.COUNT: AGG.SUM, AGG.MIN: AGG.MIN, AGG.MAX: AGG.MAX, AGG.SOME: AGG.SOME}
    out_resolver: List[Tuple[int, str]] = []
    aggs = []
    next_id = _MERGE_ID_BASE
    for a in gb.aggregates:
        name = _column_name(a.column, names)
        out_resolver.append((next_id, _PARTIAL_PREFIX + name))
        out_col = a.column if isinstance(a.column, Column) else Column(id=a.column) if isinstance(a.column, int) else Column(name=a.column)
        aggs.append(Aggregate(out_col, merge_of[a.id], [next_id]))
        next_id += 1
    keys = []
    for k in gb.keys:
        name = _column_name(k, names)
        out_resolver.append((next_id, name))
        keys.append(next_id)
        next_id += 1
    merged = Program(version=program.version, kernels=program.kernels).group_by(aggs, keys=keys)
    # the program's commands after the aggregation apply to the merged groups; table columns (the keys) have other ids
    # in the merge program's resolver, so every reference goes by name
    by_name = lambda c: Column(name=_column_name(c, names))  # noqa: E731
    _, tail = split_at_last_group_by(program)
    for cmd in tail:
        if isinstance(cmd, Projection):
            merged.commands.append(Projection([by_name(c) for c in cmd.columns]))
        elif isinstance(cmd, Filter):
            merged.commands.append(Filter(by_name(cmd.predicate)))
        elif isinstance(cmd, Assign):
            fn = cmd.function
            if fn is not None:
                fn = Function(fn.id, [by_name(a) for a in fn.args], fn.function_type, fn.kernel_idx, fn.yql_op)
            out = cmd.column if cmd.column is None or isinstance(cmd.column, Column) else Column(id=cmd.column) if isinstance(cmd.column, int) else Column(name=cmd.column)
            merged.commands.append(Assign(out, fn, cmd.constant, cmd.null, cmd.parameter, cmd.external))
        else:
            raise ValueError(f"command {type(cmd).__name__} after the last GroupBy cannot be merged across shards")
    return merged, out_resolver


def partial_batch(parts: Sequence[Dict[str, ResultColumn]], program: Program, resolver: Resolver) -> Dict[str, Col]:
    """Concatenates per-shard results column-wise into the input batch of the merge program."""
    names = {i: n for i, n in resolver}
    gb = last_group_by(program)
    agg_names = [_column_name(a.column, names) for a in gb.aggregates]
    key_names = [_column_name(k, names) for k in gb.keys]
    out: Dict[str, Col] = {}
    for name in agg_names + key_names:
        cols = [p[name] for p in parts]
        valid = np.concatenate([np.asarray(c.valid, dtype=np.bool_) for c in cols]) if cols else np.zeros(0, np.bool_)
        if isinstance(cols[0].values, list):
            vals: List[Optional[bytes]] = []
            for c in cols:
                vals.extend(v if ok else None for v, ok in zip(c.values, c.valid))
            col = Col(vals, type=cols[0].type)
        else:
            vals_np = np.concatenate([np.asarray(c.values) for c in cols])
            col = Col(vals_np, valid=None if valid.all() else valid, type=cols[0].type)
        out[(_PARTIAL_PREFIX + name) if name in agg_names else name] = col
    return out


# ---- wire format of a result batch (all-gather payload) --------------------------------------------------
def pack_result(result: Dict[str, ResultColumn]) -> bytes:
    out = [struct.pack("<I", len(result))]
    for key, c in result.items():
        name = c.name.encode()
        typ = c.type.encode()
        n = len(c.valid)
        out.append(struct.pack("<HHq", len(name), len(typ), n))
        out.append(name)
        out.append(typ)
        out.append(np.asarray(c.valid, dtype=np.uint8).tobytes())
        if isinstance(c.values, list):
            lens = np.array([len(v) for v in c.values], dtype=np.int32)
            out.append(lens.tobytes())
            out.append(b"".join(bytes(v) for v in c.values))
        else:
            out.append(np.ascontiguousarray(c.values).tobytes())
    return b"".join(out)


def unpack_result(data: bytes) -> Dict[str, ResultColumn]:
    from .arrow_c import _TYPES
    pos = 0
    (ncols,) = struct.unpack_from("<I", data, pos)
    pos += 4
    out: Dict[str, ResultColumn] = {}
    for _ in range(ncols):
        ln, lt, n = struct.unpack_from("<HHq", data, pos)
        pos += 12
        name = data[pos:pos + ln].decode()
        pos += ln
        typ = data[pos:pos + lt].decode()
        pos += lt
        valid = np.frombuffer(data, dtype=np.uint8, count=n, offset=pos).astype(np.bool_)
        pos += n
        if typ in ("string", "binary"):
            lens = np.frombuffer(data, dtype=np.int32, count=n, offset=pos)
            pos += 4 * n
            vals = []
            for l in lens:
                vals.append(data[pos:pos + int(l)])
                pos += int(l)
            col = ResultColumn(name, typ, vals, valid)
        else:
            dt = np.dtype(_TYPES[typ][1])
            vals_np = np.frombuffer(data, dtype=dt, count=n, offset=pos).copy()
            pos += n * dt.itemsize
            col = ResultColumn(name, typ, vals_np, valid)
        key = name
        while key in out:
            key += "'"
        out[key] = col
    return out


def all_gather_results(result: Dict[str, ResultColumn], device=None) -> List[Dict[str, ResultColumn]]:
    """Replicates every rank's partial table on every rank: one size all-gather, one payload all-gather."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    payload = pack_result(result)
    dev = device if device is not None else torch.device("cpu")
    size = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, size)
    cap = max(int(s.item()) for s in sizes)
    buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
    buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    bufs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(bufs, buf)
    return [unpack_result(b[:int(s.item())].cpu().numpy().tobytes()) for b, s in zip(bufs, sizes)]


def key_hashes(result: Dict[str, ResultColumn], key_names: Sequence[str]) -> np.ndarray:
    """64-bit hash of every row's key tuple (value bytes; a null hashes as its own marker), the same on every rank."""
    n = len(next(iter(result.values())).valid) if result else 0
    h = np.full(n, 0x9E3779B97F4A7C15, dtype=np.uint64)
    with np.errstate(over="ignore"):
        for name in key_names:
            c = result[name]
            if isinstance(c.values, list):
                import zlib
                v = np.array([zlib.crc32(bytes(x)) | (len(x) << 32) for x in c.values], dtype=np.uint64)
            else:
                a = np.ascontiguousarray(c.values)
                v = a.view(np.dtype(f"u{a.dtype.itemsize}")).astype(np.uint64)
            v = np.where(np.asarray(c.valid, dtype=np.bool_), v, np.uint64(0xDEADBEEFCAFEF00D))
            h = (h ^ v) * np.uint64(0xFF51AFD7ED558CCD)
            h ^= h >> np.uint64(33)
    return h


def take_rows(result: Dict[str, ResultColumn], rows: np.ndarray) -> Dict[str, ResultColumn]:
    out: Dict[str, ResultColumn] = {}
    for key, c in result.items():
        if isinstance(c.values, list):
            vals = [c.values[i] for i in rows]
        else:
            vals = np.asarray(c.values)[rows]
        out[key] = ResultColumn(c.name, c.type, vals, np.asarray(c.valid)[rows], getattr(c, "decimal", None))
    return out


def all_to_all_results(result: Dict[str, ResultColumn], key_names: Sequence[str], device=None) -> List[Dict[str, ResultColumn]]:
    """Hash-partitioned exchange for high-cardinality groupings: every group goes to the rank that owns its key hash,
    so each rank merges 1/N of the groups and the final result stays distributed.  Point-to-point sends (RCCL over
    xGMI keeps all links busy; gloo on CPU) -- one size exchange, one payload exchange."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    owner = partition_of_keys(key_hashes(result, key_names), world)
    payloads = [pack_result(take_rows(result, np.nonzero(owner == r)[0])) for r in range(world)]
    sizes_out = torch.tensor([len(p) for p in payloads], dtype=torch.int64, device=dev)
    sizes_all = [torch.zeros(world, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes_all, sizes_out)
    sizes_in = [int(sizes_all[r][rank].item()) for r in range(world)]
    send = [torch.frombuffer(bytearray(p), dtype=torch.uint8).to(dev) for p in payloads]
    recv = [torch.empty(n, dtype=torch.uint8, device=dev) for n in sizes_in]
    ops = []
    for r in range(world):
        if r == rank:
            recv[r] = send[r]
            continue
        ops.append(dist.P2POp(dist.isend, send[r], r))
        ops.append(dist.P2POp(dist.irecv, recv[r], r))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return [unpack_result(t.cpu().numpy().tobytes()) for t in recv]


def partition_of_keys(key_hashes: np.ndarray, world: int) -> np.ndarray:
    """Rank that owns each group in the hash-partitioned (all-to-all) merge of high-cardinality groupings."""
    h = np.asarray(key_hashes, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = (h ^ (h >> np.uint64(33))) * np.uint64(0xFF51AFD7ED558CCD)
        h = h ^ (h >> np.uint64(33))
    return (h % np.uint64(world)).astype(np.int64)


class ShardedScan:
    """One rank of a row-range sharded scan: run the program on the local shard, merge across ranks."""

    def __init__(self, program: Program, resolver: Resolver, device: int = 0):
        from .program import ProgramContainer

        self.program = program
        self.resolver = list(resolver)
        self.device = device
        self.local = ProgramContainer(device)
        ok, err = self.local.Init(self.resolver, program)
        if not ok:
            raise ValueError(err)
        self.merge = None
        self._head = None  # the program up to its last GroupBy: what a shard runs when there is a merge to follow
        if last_group_by(program) is not None:
            mp, mr = merge_program(program, resolver)
            self.merge_resolver = mr
            self.merge = ProgramContainer(device)
            ok, err = self.merge.Init(mr, mp)
            if not ok:
                raise ValueError(err)

    # partial tables larger than this go through the hash-partitioned exchange
    ALL_TO_ALL_GROUPS = 65536

    def run(self, batch, gather_device=None, exchange: str = "auto", keep_on_device: bool = False, lazy: bool = False):
        """Local scan (HBM-resident `batch`), then -- when torch.distributed is initialised with more than one rank --
        the exchange of the partial tables and the merge program.  exchange = "allgather": every rank ends with the
        whole result; "alltoall": every rank ends with the groups whose key hash it owns (their union is the result);
        "auto": all-to-all once any rank's partial table exceeds ALL_TO_ALL_GROUPS rows.  `gather_device`: where the
        collective's tensors live -- the GPU (RCCL), or the CPU when the process group is gloo (a rehearsal of several
        ranks on one GPU): the packed parts then cross the host on their way.  `lazy`: the result stays in the Arrow
        structs the library exported to the host (ProgramContainer.Export)."""
        import torch.distributed as dist

        single = not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        if single:
            result = self.local.ApplyDevice(batch)
            return result if keep_on_device else self.local.Export(result, lazy)  # one rank: nothing to exchange
        if self.merge is None:
            raise ValueError("sharded scan without aggregation: results stay on their shards")
        partial = self.head.ApplyDevice(batch)
        merged = self.merge.ApplyDevice(self.exchange_device(partial, gather_device, exchange))
        return merged if keep_on_device else self.merge.Export(merged, lazy)

    @property
    def head(self):
        """The executor of the program's part up to its last GroupBy (the whole program when nothing follows it)."""
        if self._head is None:
            head, tail = split_at_last_group_by(self.program)
            if not tail:
                self._head = self.local
            else:
                from .program import ProgramContainer
                self._head = ProgramContainer(self.device)
                ok, err = self._head.Init(self.resolver, head)
                if not ok:
                    raise ValueError(err)
        return self._head

    HEADER_CAPACITY = 16384  # bytes of a part's description (a few dozen per column)
    SLOT_BYTES = 65536       # a rank's contribution to the one-collective exchange of small partial tables ...
    SLOT_HEADER = 4096       # ... of which the first bytes hold three int64 (description length, data bytes, fits) and the description

    def _merge_names(self, partial) -> List[str]:
        names = {i: n for i, n in self.resolver}
        agg_names = {_column_name(a.column, names) for a in last_group_by(self.program).aggregates}
        return [(_PARTIAL_PREFIX + n) if n in agg_names else n for n in self.head.ColumnNames(partial)]

    def exchange_device(self, partial, device=None, exchange: str = "auto"):
        """The merge program's input on this rank as an HBM-resident batch: the partial tables of all ranks
        (all-gather), or the groups of all ranks whose key hash this rank owns (all-to-all)."""
        import torch
        import torch.distributed as dist

        pc = self.head
        world, rank = dist.get_world_size(), dist.get_rank()
        gpu = torch.device("cuda", torch.cuda.current_device())
        wire = device if device is not None else gpu  # where the collective's tensors live
        names = {i: n for i, n in self.resolver}
        key_names = [_column_name(k, names) for k in last_group_by(self.program).keys]
        cap = self.HEADER_CAPACITY

        def describe(parts):
            """One row of int64 per part: header length, data bytes, then the header itself."""
            t = torch.zeros((len(parts), 2 + cap // 8), dtype=torch.int64)
            for i, p in enumerate(parts):
                header, data_bytes = pc.PackLayout(p)
                if len(header) > cap:
                    raise ValueError("result batch with too many columns to exchange")
                t[i, 0], t[i, 1] = len(header), data_bytes
                t[i, 2:].view(torch.uint8)[:len(header)] = torch.frombuffer(bytearray(header), dtype=torch.uint8)
            return t

        def parse(row):
            hl, db = int(row[0]), int(row[1])
            return row[2:].view(torch.uint8)[:hl].numpy().tobytes(), db

        def rows_of(header: bytes) -> int:
            return struct.unpack_from("<q", header, 8)[0]

        import os
        import time
        timing = os.environ.get("SSA_EXCHANGE_TIMING") and rank == 0
        marks = [("start", time.perf_counter())]

        def mark(what):
            if timing:
                torch.cuda.synchronize()
                marks.append((what, time.perf_counter()))

        def settle():
            # A collective on device tensors is queued on the process group's stream, which torch's current stream then
            # waits for -- the executor's own stream does not: what it reads next must have arrived.
            if wire.type == "cuda":
                torch.cuda.current_stream(wire).synchronize()

        heads = None
        if exchange in ("auto", "allgather"):
            # A partial table of a few groups (TPC-H Q1: four rows of ten columns) goes out in ONE collective: every rank
            # contributes a slot of fixed size that carries the description of its table and the packed table itself.
            # The slot is filled on the device (the description is a few hundred bytes copied up, the table is packed in
            # place); only after the collective do the descriptions come down, all of them in one copy.  A rank whose table
            # does not fit says so in its slot, and every rank -- they all see the same slots -- takes the two-step way below.
            slot_bytes, slot_head = self.SLOT_BYTES, self.SLOT_HEADER
            header, data_bytes = pc.PackLayout(partial)
            fits = len(header) <= slot_head - 24 and data_bytes <= slot_bytes - slot_head
            slot = torch.empty(slot_bytes, dtype=torch.uint8, device=gpu)
            lead = struct.pack("<qqq", len(header), data_bytes, 1 if fits else 0) + (header if fits else b"")
            slot[:len(lead)].copy_(torch.frombuffer(bytearray(lead), dtype=torch.uint8), non_blocking=True)
            if fits:
                pc.PackData(partial, slot.data_ptr() + slot_head, data_bytes)
            slots = torch.empty(world * slot_bytes, dtype=torch.uint8, device=wire)
            dist.all_gather_into_tensor(slots, slot.to(wire))
            settle()
            slots = slots.to(gpu)
            leads = slots.view(world, slot_bytes)[:, :slot_head].cpu().numpy()
            mark("one collective")
            seen = [struct.unpack_from("<qqq", leads[r].tobytes(), 0) for r in range(world)]
            if all(f == 1 for _, _, f in seen):
                heads = [(leads[r, 24:24 + seen[r][0]].tobytes(), seen[r][1]) for r in range(world)]
                most = max(rows_of(h) for h, _ in heads)
                if exchange == "allgather" or not (key_names and most > self.ALL_TO_ALL_GROUPS):
                    parts = [pc.Unpack(h, slots.data_ptr() + r * slot_bytes + slot_head, db, keep=slots) for r, (h, db) in enumerate(heads)]
                    whole = pc.Concat(parts)
                    mark("concatenate")
                    if timing:
                        print("exchange: " + ", ".join(f"{w} {1e3 * (t - marks[i][1]):.2f} ms" for i, (w, t) in enumerate(marks[1:])), flush=True)
                    return pc.Rename(whole, self._merge_names(whole))
                exchange = "alltoall"
                heads = None
        if heads is None and exchange in ("auto", "allgather"):
            # every rank's description of its whole partial table: enough to decide, and the first half of an all-gather
            mine = describe([partial]).reshape(-1).to(wire)
            every = torch.empty(world * mine.numel(), dtype=torch.int64, device=wire)
            dist.all_gather_into_tensor(every, mine)
            heads = [parse(r) for r in every.cpu().reshape(world, -1)]
            if exchange == "auto":
                most = max(rows_of(h) for h, _ in heads)
                exchange = "alltoall" if (key_names and most > self.ALL_TO_ALL_GROUPS) else "allgather"
            mark("descriptions")

        if exchange == "allgather":
            width = max(db for _, db in heads)
            send = torch.empty(width, dtype=torch.uint8, device=gpu)
            pc.PackData(partial, send.data_ptr(), width)
            recv = torch.empty(world * width, dtype=torch.uint8, device=wire)
            dist.all_gather_into_tensor(recv, send.to(wire))
            settle()
            recv = recv.to(gpu)
            mark("all-gather")
            parts = [pc.Unpack(h, recv.data_ptr() + r * width, db, keep=recv) for r, (h, db) in enumerate(heads)]
        else:
            # (by the key hash alone: the reference's split also balances its lists, which sends equal keys to different places)
            split, counts = pc.PartitionByKey(partial, key_names, world)
            mark("partition by key hash")
            pieces, at = [], 0
            for r in range(world):
                pieces.append(pc.Slice(split, at, counts[r]))
                at += counts[r]
            mine = describe(pieces).reshape(-1).to(wire)  # part r: what rank r gets from this rank
            theirs = torch.empty_like(mine)               # part r: what this rank gets from rank r
            dist.all_to_all_single(theirs, mine)
            out_heads = [parse(r) for r in mine.cpu().reshape(world, -1)]
            in_heads = [parse(r) for r in theirs.cpu().reshape(world, -1)]
            out_sizes = [db for _, db in out_heads]
            in_sizes = [db for _, db in in_heads]
            send = torch.empty(sum(out_sizes), dtype=torch.uint8, device=gpu)
            at = 0
            for r in range(world):
                pc.PackData(pieces[r], send.data_ptr() + at, out_sizes[r])
                at += out_sizes[r]
            mark("pack")
            recv = torch.empty(sum(in_sizes), dtype=torch.uint8, device=wire)
            dist.all_to_all_single(recv, send.to(wire), output_split_sizes=in_sizes, input_split_sizes=out_sizes)
            settle()
            recv = recv.to(gpu)
            mark("all-to-all")
            parts, at = [], 0
            for r in range(world):
                parts.append(pc.Unpack(in_heads[r][0], recv.data_ptr() + at, in_sizes[r], keep=recv))
                at += in_sizes[r]
        whole = pc.Concat(parts)
        mark("concatenate")
        if timing:
            print("exchange: " + ", ".join(f"{w} {1e3 * (t - marks[i][1]):.2f} ms" for i, (w, t) in enumerate(marks[1:])), flush=True)
        return pc.Rename(whole, self._merge_names(whole))

    def exchange(self, partial: Dict[str, ResultColumn], device=None, exchange: str = "auto") -> Dict[str, Col]:
        """The merge program's input batch on this rank."""
        import torch
        import torch.distributed as dist

        names = {i: n for i, n in self.resolver}
        key_names = [_column_name(k, names) for k in last_group_by(self.program).keys]
        if exchange == "auto":
            n = torch.tensor([len(next(iter(partial.values())).valid)], dtype=torch.int64, device=device or torch.device("cpu"))
            dist.all_reduce(n, op=dist.ReduceOp.MAX)
            exchange = "alltoall" if (key_names and int(n.item()) > self.ALL_TO_ALL_GROUPS) else "allgather"
        parts = all_to_all_results(partial, key_names, device) if exchange == "alltoall" else all_gather_results(partial, device)
        return partial_batch(parts, self.program, self.resolver)
