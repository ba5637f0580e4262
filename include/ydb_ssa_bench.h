/*
 * C ABI of libydb_ssa_bench.so: benchmark and test support for the MI355X executor of YDB column-shard SSA programs.
 * Nothing here replaces an interface of the reference; the executor's own library (ydb_ssa_hip.h) does not contain it.
 */
#ifndef YDB_SSA_BENCH_H
#define YDB_SSA_BENCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Fills caller-provided device arrays with rows [row0, row0 + nrows) of the deterministic synthetic lineitem
 * (counter-based splitmix64 of `seed`, column id and row number).  Any pointer may be NULL to skip that column. */
struct ssa_lineitem_columns {
    int32_t* l_shipdate;      /* days since epoch */
    double* l_quantity;
    double* l_extendedprice;
    double* l_discount;
    double* l_tax;
    int32_t* rf_offsets;      /* l_returnflag utf8: nrows + 1 offsets */
    uint8_t* rf_chars;
    int32_t* ls_offsets;      /* l_linestatus utf8 */
    uint8_t* ls_chars;
    int8_t* rf_code;          /* dictionary codes: 0 = "A", 1 = "N", 2 = "R" */
    int8_t* ls_code;          /* 0 = "F", 1 = "O" */
};
int ssa_gen_lineitem(const struct ssa_lineitem_columns* cols, int64_t row0, int64_t nrows, uint64_t seed, int device);

#ifdef __cplusplus
}
#endif

#endif /* YDB_SSA_BENCH_H */
