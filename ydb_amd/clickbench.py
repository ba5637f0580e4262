"""ClickBench `hits` workloads of the scan-pushdown path: the programs and a synthetic table resident in HBM.

Programs are the `SsaProgram` nodes of the reference's canonised column-table plans
(ydb/tests/functional/clickbench/canondata/test.test_plans_column_/queries-original-plan-column-{0,1,2,3,6}) plus the
group-bys that KQP keeps above the scan at this snapshot (ydb/core/kqp/opt/physical/kqp_opt_phy_olap_agg.cpp:302-305),
written as SSA GroupBy commands: a small int16 key (AdvEngineID), a high-cardinality int64 key (UserID) and a string key
(SearchPhrase).  Column types follow ydb/library/workload/clickbench/click_bench_schema.sql:6-46.

The table is synthetic (the reference's hits sample is not in its tree): the cardinalities and skews are the public
facts of the dataset -- AdvEngineID is 0 in about 97 % of rows, about 17.6 M distinct UserID and 6 M distinct
SearchPhrase per 100 M rows, 87 % of the phrases empty -- and only parameterise the generator.  torch fills the columns
on the device; it is plumbing here (device memory and a random generator), the scans are the library's.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

from .ssa_pb import AGG, YQL, Aggregate, Program

HITS_ROWS = 99_997_497
HITS_RESOLVER: List[Tuple[int, str]] = [(6, "EventDate"), (10, "UserID"), (14, "URL"), (21, "ResolutionWidth"), (40, "SearchPhrase"),
                                        (41, "AdvEngineID")]


def programs() -> Dict[str, Program]:
    p: Dict[str, Program] = {}
    # Q1: SELECT COUNT(*) FROM hits WHERE AdvEngineID <> 0
    p["cb_q1"] = (Program(version=5).assign_const(106, "Int32", 0).assign_yql(107, YQL.NotEquals, [41, 106]).filter(107)
                  .group_by([Aggregate(108, AGG.COUNT, [])]).project([108]))
    # Q2: SELECT SUM(AdvEngineID), COUNT(*), AVG(ResolutionWidth)
    p["cb_q2"] = (Program(version=5).group_by([Aggregate(106, AGG.SUM, [21]), Aggregate(107, AGG.COUNT, [21]), Aggregate(108, AGG.COUNT, []),
                                                Aggregate(109, AGG.SUM, [41])]).project([107, 106, 108, 109]))
    # Q3: SELECT AVG(UserID)
    p["cb_q3"] = Program(version=5).group_by([Aggregate(106, AGG.SUM, [10]), Aggregate(107, AGG.COUNT, [10])]).project([107, 106])
    # Q6: SELECT MIN(EventDate), MAX(EventDate)
    p["cb_q6"] = Program(version=5).group_by([Aggregate(106, AGG.MAX, [6]), Aggregate(107, AGG.MIN, [6])]).project([106, 107])
    # Q20: SELECT COUNT(*) FROM hits WHERE URL LIKE '%google%'
    p["cb_q20"] = (Program(version=5).assign_const(106, "Bytes", b"google").assign_yql(107, YQL.StringContains, [14, 106]).filter(107)
                   .group_by([Aggregate(108, AGG.COUNT, [])]).project([108]))
    # Q7: SELECT AdvEngineID, COUNT(*) WHERE AdvEngineID <> 0 GROUP BY AdvEngineID
    p["cb_adv"] = (Program().assign_const(106, "Int32", 0).assign_yql(107, YQL.NotEquals, [41, 106]).filter(107)
                   .group_by([Aggregate(108, AGG.COUNT, [])], keys=[41]))
    # Q15-like: SELECT UserID, COUNT(*) GROUP BY UserID
    p["cb_userid"] = Program().group_by([Aggregate(108, AGG.COUNT, [])], keys=[10])
    # Q12-like: SELECT SearchPhrase, COUNT(*) GROUP BY SearchPhrase
    p["cb_phrase"] = Program().group_by([Aggregate(108, AGG.COUNT, [])], keys=[40])
    return p


# source columns each program reads and their algorithmic bytes per row (values, plus 4-byte offsets for the utf8 key;
# the phrase bytes are added from the generated table)
COLUMNS = {
    "cb_q1": (["AdvEngineID"], 2), "cb_q2": (["ResolutionWidth", "AdvEngineID"], 4), "cb_q3": (["UserID"], 8),
    "cb_q6": (["EventDate"], 4), "cb_adv": (["AdvEngineID"], 2), "cb_userid": (["UserID"], 8), "cb_phrase": (["SearchPhrase"], 4),
    "cb_q20": (["URL"], 4),
}


class DeviceHits:
    """Columns of a synthetic `hits` in HBM.  `distinct_users` / `distinct_phrases` scale with the row count."""

    def __init__(self, nrows: int = HITS_ROWS, device: int = 0, seed: int = 0x5EED0002, row0: int = 0, columns=None):
        import torch
        dev = torch.device("cuda", device)
        g = torch.Generator(device=dev)
        g.manual_seed(seed + row0)
        self.nrows = nrows
        self.t = {}
        want = set(columns) if columns is not None else {"EventDate", "UserID", "ResolutionWidth", "SearchPhrase", "AdvEngineID", "URL"}
        if "AdvEngineID" in want:
            u = torch.rand(nrows, device=dev, generator=g)
            adv = torch.randint(1, 31, (nrows,), device=dev, generator=g, dtype=torch.int16)
            self.t["AdvEngineID"] = torch.where(u < 0.97, torch.zeros_like(adv), adv)
        if "ResolutionWidth" in want:
            widths = torch.tensor([0, 1024, 1280, 1366, 1440, 1600, 1920], device=dev, dtype=torch.int16)
            self.t["ResolutionWidth"] = widths[torch.randint(0, 7, (nrows,), device=dev, generator=g)]
        if "EventDate" in want:
            self.t["EventDate"] = (15887 + torch.randint(0, 31, (nrows,), device=dev, generator=g, dtype=torch.int32))
        if "UserID" in want:
            # zipf-like ranks over about 17.6 % of the rows, spread over the int64 range by an odd multiplier
            distinct = max(1, int(nrows * 0.176))
            u = torch.rand(nrows, device=dev, generator=g, dtype=torch.float64)
            ranks = (distinct ** u).to(torch.int64) - 1          # log-uniform: heavy head, long tail
            ranks = torch.where(torch.rand(nrows, device=dev, generator=g) < 0.85,
                                torch.randint(0, distinct, (nrows,), device=dev, generator=g), ranks)
            self.t["UserID"] = ranks * (0x9E3779B97F4A7C15 - 2**64) + 12345   # the multiplier as a signed 64-bit value; wraps
            del ranks, u
        if "SearchPhrase" in want:
            distinct = max(1, int(nrows * 0.06))
            empty = torch.rand(nrows, device=dev, generator=g) < 0.87
            ids = torch.randint(0, distinct, (nrows,), device=dev, generator=g)
            lens = 9 + (ids * 2654435761 >> 7) % 33                      # 9..41 bytes, mean 25, fixed per phrase
            lens = torch.where(empty, torch.zeros_like(lens), lens)
            offsets = torch.zeros(nrows + 1, dtype=torch.int64, device=dev)
            torch.cumsum(lens, 0, out=offsets[1:])
            total = int(offsets[-1].item())
            if total >= 2**31:
                raise ValueError("SearchPhrase bytes exceed int32 offsets: use fewer rows per batch")
            row_of = torch.repeat_interleave(torch.arange(nrows, device=dev), lens, output_size=total)
            pos = torch.arange(total, device=dev) - offsets[row_of]
            h = (ids[row_of] * 1000003 + pos * 7919)
            h = (h ^ (h >> 13)) * 0x5851F42D4C957F2D
            chars = (97 + ((h >> 40) % 26)).to(torch.uint8)
            self.t["SearchPhrase_offsets"] = offsets.to(torch.int32)
            self.t["SearchPhrase_chars"] = torch.cat([chars, torch.zeros(16, dtype=torch.uint8, device=dev)])
            self.phrase_bytes = total
            del row_of, pos, h, chars, offsets, lens, ids, empty
        if "URL" in want:
            # 20..119 bytes (mean about 70) of lower-case letters, '/', '.'; one URL in 64 carries "google" somewhere
            lens = 20 + torch.randint(0, 100, (nrows,), device=dev, generator=g)
            offsets = torch.zeros(nrows + 1, dtype=torch.int64, device=dev)
            torch.cumsum(lens, 0, out=offsets[1:])
            total = int(offsets[-1].item())
            if total >= 2**31:
                raise ValueError("URL bytes exceed int32 offsets: use fewer rows per batch")
            chars = torch.randint(0, 28, (total + 16,), device=dev, generator=g, dtype=torch.uint8)
            chars = torch.where(chars < 26, chars + 97, torch.where(chars == 26, torch.full_like(chars, 47), torch.full_like(chars, 46)))
            chars[total:] = 0
            marked = torch.nonzero(torch.randint(0, 64, (nrows,), device=dev, generator=g) == 0).flatten()
            at = offsets[marked] + (torch.randint(0, 2**30, (len(marked),), device=dev, generator=g) % (lens[marked] - 6))
            word = torch.tensor(list(b"google"), dtype=torch.uint8, device=dev)
            for j in range(6):
                chars[at + j] = word[j]
            self.t["URL_offsets"] = offsets.to(torch.int32)
            self.t["URL_chars"] = chars
            self.url_bytes = total
            del lens, offsets, marked, at
        torch.cuda.synchronize(dev)

    def batch(self, columns):
        from .program import device_batch
        T = {"int16": 3, "int32": 4, "int64": 5, "string": 14}
        desc = []
        for name in columns:
            if name in ("AdvEngineID", "ResolutionWidth"):
                desc.append((name, T["int16"], self.t[name].data_ptr(), 0, 0, 0, 0))
            elif name == "EventDate":
                desc.append((name, T["int32"], self.t[name].data_ptr(), 0, 0, 0, 0))
            elif name == "UserID":
                desc.append((name, T["int64"], self.t[name].data_ptr(), 0, 0, 0, 0))
            elif name == "SearchPhrase":
                desc.append((name, T["string"], self.t["SearchPhrase_offsets"].data_ptr(), 0, self.t["SearchPhrase_chars"].data_ptr(),
                             self.phrase_bytes, 0))
            elif name == "URL":
                desc.append((name, T["string"], self.t["URL_offsets"].data_ptr(), 0, self.t["URL_chars"].data_ptr(), self.url_bytes, 0))
            else:
                raise KeyError(name)
        return device_batch(desc, self.nrows, keep=self)

    def host(self, columns) -> Dict[str, object]:
        """The same columns as numpy arrays / lists of str (for the oracle)."""
        import numpy as np
        out = {}
        for name in columns:
            if name in ("SearchPhrase", "URL"):
                offs = self.t[name + "_offsets"].cpu().numpy()
                chars = self.t[name + "_chars"].cpu().numpy().tobytes()
                out[name] = [chars[offs[i]:offs[i + 1]].decode() for i in range(self.nrows)]
            else:
                out[name] = np.ascontiguousarray(self.t[name].cpu().numpy())
        return out
