// libydb_ssa_bench.so: the synthetic TPC-H lineitem of the benchmark and the parity tests, generated in HBM.  Not part of
// the executor's library: nothing here replaces an interface of the reference.
#include <hip/hip_runtime.h>

#include "../../include/ydb_ssa_bench.h"
#include "kernels.h"

extern "C" int ssa_gen_lineitem(const ssa_lineitem_columns* cols, int64_t row0, int64_t nrows, uint64_t seed, int device) {
    if (cols == nullptr || nrows < 0) return 1;
    if (hipSetDevice(device) != hipSuccess) return 4;
    ssa::LineitemColumns c{};
    c.l_shipdate = cols->l_shipdate;
    c.l_quantity = cols->l_quantity;
    c.l_extendedprice = cols->l_extendedprice;
    c.l_discount = cols->l_discount;
    c.l_tax = cols->l_tax;
    c.rf_offsets = cols->rf_offsets;
    c.rf_chars = cols->rf_chars;
    c.ls_offsets = cols->ls_offsets;
    c.ls_chars = cols->ls_chars;
    c.rf_code = cols->rf_code;
    c.ls_code = cols->ls_code;
    if (ssa::LaunchGenLineitem(c, row0, nrows, seed, nullptr) != hipSuccess) return 4;
    if (hipDeviceSynchronize() != hipSuccess) return 4;
    return 0;
}
