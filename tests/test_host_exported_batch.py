"""ExportedBatch: the Arrow structs of an exported batch read as the dictionary read_batch makes of them, converted when
first looked at (no GPU: the structs come from HostBatch, the host-side builder of the import path)."""
import numpy as np

from ydb_amd.arrow_c import Col, ExportedBatch, HostBatch, read_batch


def test_lazy_batch_reads_like_the_eager_one():
    hb = HostBatch({"a": np.arange(5, dtype=np.int64), "s": Col([b"x", b"", None, b"q", b"long one"], type="string"),
                    "f": Col(np.linspace(0, 1, 5), valid=np.array([1, 0, 1, 1, 1], dtype=np.bool_))})
    eager = read_batch(hb.array, hb.schema, release=False)
    lazy = ExportedBatch(hb.array, hb.schema)
    assert lazy.rows == 5
    assert lazy._cols is None          # nothing converted yet
    assert list(lazy) == list(eager) and len(lazy) == 3
    for name in eager:
        assert np.array_equal(np.asarray(lazy[name].valid), np.asarray(eager[name].valid))
        assert lazy[name].to_pylist() == eager[name].to_pylist()
    assert lazy.rows == 5              # and after the conversion
