"""TPC-H Q1 / Q6 as hand-built SSA programs over a synthetic lineitem, plus the data generator.

The reference's KQP layer does not push keyed GROUP BY or sum(a*b) down at this snapshot
(ydb/core/kqp/opt/physical/kqp_opt_phy_olap_agg.cpp:44-54,302-305), so -- like the reference's own
unit tests (ydb/core/tx/columnshard/ut_rw/ut_columnshard_read_write.cpp:1108-1223) -- the programs
are assembled directly against ssa.proto.  Query text: ydb/library/benchmarks/queries/tpch/ydb/q1.sql:5-25
and q6.sql:5-15; column types: ydb/library/workload/tpch/tpch_schema.sql:19-37 (Date32 -> int32 days,
Double money columns, Utf8 flags).
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Tuple

import numpy as np

from .ssa_pb import AGG, F, Aggregate, Program

# column ids of the lineitem column table (resolver)
L_QUANTITY, L_EXTENDEDPRICE, L_DISCOUNT, L_TAX, L_RETURNFLAG, L_LINESTATUS, L_SHIPDATE = 5, 6, 7, 8, 9, 10, 11
L_RF_CODE, L_LS_CODE = 17, 18  # dictionary-coded flag columns (1-byte codes) of the "codes" variant

LINEITEM_RESOLVER = [
    (L_SHIPDATE, "l_shipdate"), (L_QUANTITY, "l_quantity"), (L_EXTENDEDPRICE, "l_extendedprice"),
    (L_DISCOUNT, "l_discount"), (L_TAX, "l_tax"), (L_RETURNFLAG, "l_returnflag"), (L_LINESTATUS, "l_linestatus"),
    (L_RF_CODE, "l_returnflag_code"), (L_LS_CODE, "l_linestatus_code"),
]

SEED_LINEITEM = 0x5EED0001
SF1_ROWS = 6_001_215
SF100_ROWS = 600_037_902

Q6_DATE_LO = 9131          # 1995-01-01 in days since epoch
Q6_DATE_HI = 9131 + 365    # + 1 year
Q1_DATE = 10461            # 1998-12-01 - 100 days = 1998-08-23

Q6_BYTES_PER_ROW = 4 + 3 * 8                    # l_shipdate + quantity, extendedprice, discount
Q1_BYTES_PER_ROW_UTF8 = 4 + 4 * 8 + 2 * (4 + 1)  # + tax, two utf8 flags (offset + 1 char each)
Q1_BYTES_PER_ROW_CODES = 4 + 4 * 8 + 2 * 1       # flags as 1-byte dictionary codes


def q6_program(date_lo: int = None, date_hi: int = None, discount: float = 0.07, quantity: float = 25.0) -> Program:
    """select sum(l_extendedprice * l_discount) where shipdate in [d0, d0 + 1y) and discount between .06 and .08
    and quantity < 25 -- five Filters in one step, one Assign, GroupBy{SUM}, Projection.  The arguments are the query's
    substitution parameters (the defaults: the validation values of the TPC-H specification)."""
    date_lo = Q6_DATE_LO if date_lo is None else date_lo
    date_hi = Q6_DATE_HI if date_hi is None else date_hi
    p = Program(version=5)
    p.assign_const(100, "Int32", date_lo).assign_const(101, "Int32", date_hi)
    p.assign_const(102, "Double", discount - 0.01).assign_const(103, "Double", discount + 0.01).assign_const(104, "Double", quantity)
    p.assign_func(110, F.CMP_GREATER_EQUAL, [L_SHIPDATE, 100]).assign_func(111, F.CMP_LESS, [L_SHIPDATE, 101])
    p.assign_func(112, F.CMP_GREATER_EQUAL, [L_DISCOUNT, 102]).assign_func(113, F.CMP_LESS_EQUAL, [L_DISCOUNT, 103])
    p.assign_func(114, F.CMP_LESS, [L_QUANTITY, 104])
    for c in (110, 111, 112, 113, 114):
        p.filter(c)
    p.assign_func(120, F.MATH_MULTIPLY, [L_EXTENDEDPRICE, L_DISCOUNT])
    p.group_by([Aggregate(130, AGG.SUM, [120])])
    p.project([130])
    return p


def q1_program(codes: bool = False) -> Program:
    """Q1: filter on l_shipdate, four SUMs (two over computed expressions), AVG decomposed into SUM + COUNT
    (kqp_opt_phy_olap_agg.cpp:321-334), COUNT(*), grouped by (l_returnflag, l_linestatus)."""
    rf, ls = (L_RF_CODE, L_LS_CODE) if codes else (L_RETURNFLAG, L_LINESTATUS)
    p = Program(version=5)
    p.assign_const(100, "Int32", Q1_DATE).assign_const(101, "Double", 1.0)
    p.assign_func(110, F.CMP_LESS_EQUAL, [L_SHIPDATE, 100]).filter(110)
    p.assign_func(120, F.MATH_SUBTRACT, [101, L_DISCOUNT])       # 1 - l_discount
    p.assign_func(121, F.MATH_MULTIPLY, [L_EXTENDEDPRICE, 120])  # disc_price
    p.assign_func(122, F.MATH_ADD, [101, L_TAX])                 # 1 + l_tax
    p.assign_func(123, F.MATH_MULTIPLY, [121, 122])              # charge
    p.group_by([
        Aggregate(130, AGG.SUM, [L_QUANTITY]), Aggregate(131, AGG.SUM, [L_EXTENDEDPRICE]),
        Aggregate(132, AGG.SUM, [121]), Aggregate(133, AGG.SUM, [123]),
        Aggregate(134, AGG.COUNT, [L_QUANTITY]), Aggregate(135, AGG.COUNT, [L_EXTENDEDPRICE]),
        Aggregate(136, AGG.SUM, [L_DISCOUNT]), Aggregate(137, AGG.COUNT, [L_DISCOUNT]),
        Aggregate(138, AGG.COUNT, []),
    ], keys=[rf, ls])
    return p


# ---- synthetic lineitem: the same function of (seed, column, row) on host and device -------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _draw(seed: int, col: int, rows: np.ndarray) -> np.ndarray:
    # upper 32 bits of the mixed word (every range reduction is a 32-bit modulo, as on the device)
    return _splitmix64(np.uint64(seed) ^ (np.uint64(col) << np.uint64(56)) ^ rows) >> np.uint64(32)


def lineitem_host(row0: int, nrows: int, seed: int = SEED_LINEITEM) -> Dict[str, object]:
    """numpy restatement of the device generator (gen_kernels.hip): rows [row0, row0 + nrows)."""
    rows = np.arange(row0, row0 + nrows, dtype=np.uint64)
    qty = 1 + (_draw(seed, 1, rows) % np.uint64(50)).astype(np.int64)
    cents = 90000 + (_draw(seed, 4, rows) % np.uint64(20001)).astype(np.int64) + 100 * (_draw(seed, 5, rows) % np.uint64(1000)).astype(np.int64)
    ship = (8036 + (_draw(seed, 6, rows) % np.uint64(2526)).astype(np.int64)).astype(np.int32)
    receipt = ship.astype(np.int64) + 1 + (_draw(seed, 7, rows) % np.uint64(30)).astype(np.int64)
    cutoff = 9298
    coin = (_draw(seed, 8, rows) & np.uint64(1)).astype(np.int64)
    rf = np.where(receipt <= cutoff, np.where(coin == 1, 0, 2), 1).astype(np.int8)
    ls = np.where(ship <= cutoff, 0, 1).astype(np.int8)
    return {
        "l_shipdate": ship,
        "l_quantity": qty.astype(np.float64),
        "l_extendedprice": (qty * cents).astype(np.float64) / 100.0,
        "l_discount": (_draw(seed, 2, rows) % np.uint64(11)).astype(np.float64) / 100.0,
        "l_tax": (_draw(seed, 3, rows) % np.uint64(9)).astype(np.float64) / 100.0,
        "l_returnflag": [("A", "N", "R")[c] for c in rf],
        "l_linestatus": [("F", "O")[c] for c in ls],
        "l_returnflag_code": rf,
        "l_linestatus_code": ls,
    }


class DeviceLineitem:
    """Synthetic lineitem columns resident in HBM (torch owns the memory, the library's generator fills it)."""

    def __init__(self, nrows: int, device: int = 0, seed: int = SEED_LINEITEM, row0: int = 0,
                 utf8_flags: bool = True, code_flags: bool = True):
        import torch
        from . import _lib
        lib = _lib.load()
        dev = torch.device("cuda", device)
        self.nrows = nrows
        self.t = {
            "l_shipdate": torch.empty(nrows, dtype=torch.int32, device=dev),
            "l_quantity": torch.empty(nrows, dtype=torch.float64, device=dev),
            "l_extendedprice": torch.empty(nrows, dtype=torch.float64, device=dev),
            "l_discount": torch.empty(nrows, dtype=torch.float64, device=dev),
            "l_tax": torch.empty(nrows, dtype=torch.float64, device=dev),
        }
        if utf8_flags:
            self.t["rf_offsets"] = torch.empty(nrows + 1, dtype=torch.int32, device=dev)
            self.t["rf_chars"] = torch.empty(nrows + 16, dtype=torch.uint8, device=dev)
            self.t["ls_offsets"] = torch.empty(nrows + 1, dtype=torch.int32, device=dev)
            self.t["ls_chars"] = torch.empty(nrows + 16, dtype=torch.uint8, device=dev)
        if code_flags:
            self.t["rf_code"] = torch.empty(nrows, dtype=torch.int8, device=dev)
            self.t["ls_code"] = torch.empty(nrows, dtype=torch.int8, device=dev)
        cols = _lib.LineitemColumns()
        for name in ("l_shipdate", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "rf_offsets", "rf_chars",
                     "ls_offsets", "ls_chars", "rf_code", "ls_code"):
            setattr(cols, name, self.t[name].data_ptr() if name in self.t else None)
        torch.cuda.synchronize(dev)
        rc = _lib.load_bench().ssa_gen_lineitem(ctypes.byref(cols), row0, nrows, seed, device)
        if rc != 0:
            raise RuntimeError(f"ssa_gen_lineitem failed: {rc}")

    def batch(self, columns, row0: int = 0, rows: int = None):
        """Wraps the named logical columns as an HBM-resident record batch (no copy); `row0` / `rows`: a slice of them
        (row0 a multiple of 4 keeps every column 16-byte aligned)."""
        from . import _lib
        from .program import device_batch
        T = {"int32": 4, "double": 11, "int8": 2, "string": 14}
        n = self.nrows - row0 if rows is None else rows
        assert 0 <= row0 and row0 + n <= self.nrows
        desc = []
        for name in columns:
            if name == "l_shipdate":
                desc.append((name, T["int32"], self.t[name].data_ptr() + 4 * row0, 0, 0, 0, 0))
            elif name in ("l_quantity", "l_extendedprice", "l_discount", "l_tax"):
                desc.append((name, T["double"], self.t[name].data_ptr() + 8 * row0, 0, 0, 0, 0))
            elif name == "l_returnflag":
                desc.append((name, T["string"], self.t["rf_offsets"].data_ptr() + 4 * row0, 0, self.t["rf_chars"].data_ptr(), self.nrows, 0))
            elif name == "l_linestatus":
                desc.append((name, T["string"], self.t["ls_offsets"].data_ptr() + 4 * row0, 0, self.t["ls_chars"].data_ptr(), self.nrows, 0))
            elif name in ("l_returnflag_dict", "l_linestatus_dict"):
                # the flag as a dictionary-encoded string column: the 1-byte codes the generator wrote + their letters
                import torch
                rf = name == "l_returnflag_dict"
                letters = b"ANR" if rf else b"FO"
                key = "rf_dict" if rf else "ls_dict"
                if key not in self.t:
                    dev = self.t["l_shipdate"].device
                    self.t[key] = (torch.tensor(list(letters) + [0] * 16, dtype=torch.uint8, device=dev),
                                   torch.arange(len(letters) + 1, dtype=torch.int32, device=dev))
                    torch.cuda.synchronize(dev)
                chars, offs = self.t[key]
                desc.append((name[:-5], T["string"], self.t["rf_code" if rf else "ls_code"].data_ptr() + row0, 0, chars.data_ptr(),
                             len(letters), 0, T["int8"], offs.data_ptr(), len(letters)))
            elif name == "l_returnflag_code":
                desc.append((name, T["int8"], self.t["rf_code"].data_ptr() + row0, 0, 0, 0, 0))
            elif name == "l_linestatus_code":
                desc.append((name, T["int8"], self.t["ls_code"].data_ptr() + row0, 0, 0, 0, 0))
            else:
                raise KeyError(name)
        return device_batch(desc, n, keep=self)


Q6_COLUMNS = ["l_shipdate", "l_discount", "l_quantity", "l_extendedprice"]
Q1_COLUMNS_UTF8 = ["l_shipdate", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus"]
Q1_COLUMNS_DICT = ["l_shipdate", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag_dict", "l_linestatus_dict"]
Q1_COLUMNS_CODES = ["l_shipdate", "l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag_code", "l_linestatus_code"]
