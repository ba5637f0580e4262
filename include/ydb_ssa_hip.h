/*
 * C ABI of the MI355X executor for YDB column-shard SSA programs (libydb_ssa_hip.so).
 *
 * An entry point either replaces one interface of the reference's CPU path -- the cited file:line is the reference
 * code a host shim would stop calling -- or says in its comment that it has no counterpart there and what it is for
 * (set-up and shutdown, counters and profiles for operators and tests, the exchange between GPUs, the entry point that
 * allocates nothing).  Plain pointers and sizes only: batches
 * cross the boundary through the Arrow C Data Interface (struct ArrowArray / ArrowSchema as in
 * contrib/libs/apache/arrow/cpp/src/arrow/c/abi.h), programs as the serialized protobuf a
 * ColumnShard receives.  All functions are thread-safe unless stated otherwise; none aborts.
 */
#ifndef YDB_SSA_HIP_H
#define YDB_SSA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

struct ArrowArray;
struct ArrowSchema;

typedef struct ssa_program ssa_program_t; /* parsed program + executor (one HIP stream) */
typedef struct ssa_batch ssa_batch_t;     /* record batch resident in HBM */

/* Return codes.  SSA_UNSUPPORTED means "valid program, not executable on the GPU path": the shim
 * falls back to the CPU NSsa::TProgram::ApplyTo.  SSA_INVALID corresponds to a non-OK arrow::Status
 * from the reference path (program.h:435-447), SSA_PARSE_ERROR to TProgramContainer::Init == false. */
enum ssa_status {
    SSA_OK = 0,
    SSA_INVALID = 1,
    SSA_PARSE_ERROR = 2,
    SSA_UNSUPPORTED = 3,
    SSA_HIP_ERROR = 4
};

/* Arrow type ids used by ssa_device_column (the column types a ColumnShard produces:
 * ydb/core/formats/arrow/switch/switch_type.h:125-172). */
enum ssa_type {
    SSA_T_BOOL = 1, SSA_T_INT8 = 2, SSA_T_INT16 = 3, SSA_T_INT32 = 4, SSA_T_INT64 = 5,
    SSA_T_UINT8 = 6, SSA_T_UINT16 = 7, SSA_T_UINT32 = 8, SSA_T_UINT64 = 9,
    SSA_T_FLOAT = 10, SSA_T_DOUBLE = 11, SSA_T_TIMESTAMP_US = 12, SSA_T_DURATION_US = 13,
    SSA_T_STRING = 14, SSA_T_BINARY = 15, SSA_T_DECIMAL128 = 16
};

enum { SSA_PROGRAM_OLAP_WRAPPER = 1, SSA_PROGRAM_DIRECT_OPERATIONS = 2, SSA_DIRECT_OPERATION_BASE = 1000 };

/* Library / device set-up.  Returns the number of visible HIP devices or a negative ssa_status. */
int ssa_device_count(void);
const char* ssa_version(void);
/* Ends the background builds of scan kernels: a build in progress is waited for, later scans run the kernels the
 * library was built with.  Call it before the process exits when scans of small batches may have started a build in the
 * last seconds (the compiler must not be at work while the process tears its libraries down); the Python mirror does
 * so from an atexit hook.  No counterpart in the reference. */
void ssa_shutdown(void);
/* Frees a message returned through an `err` out-parameter. */
void ssa_free_string(char* s);

/*
 * Parse a program.  Replaces NOlap::TProgramContainer::Init (ydb/core/tx/program/program.cpp:481-551):
 *  - `program` / `program_size`: serialized NKikimrSSA::TOlapProgram when bit 0 of `olap_wrapper` is set
 *    (EOlapProgramType::OLAP_PROGRAM_SSA_PROGRAM_WITH_PARAMETERS), else a bare NKikimrSSA::TProgram.
 *    With SSA_PROGRAM_DIRECT_OPERATIONS or-ed into `olap_wrapper`, a TFunction.Id of
 *    SSA_DIRECT_OPERATION_BASE + NKikimr::NKernels::EOperation (ydb/library/arrow_kernels/operations.h:5-84) stands
 *    for TAssign(column, operation, arguments) (ydb/core/formats/arrow/program.h:139-147): the operations the
 *    protobuf has no id for (gcd lcm mod modOrZero abs negate is_valid the checked arithmetic, the rounding family,
 *    sqrt e pi and the math-library functions) reach a step only through such objects (ut/ut_program_step.cpp:30-41).
 *  - `col_ids` / `col_names` / `ncols`: the IColumnResolver (ydb/core/tx/program/program.h:17-25); the
 *    first entry doubles as GetDefaultColumn() for `SELECT count(*)`.
 *  - `params` / `params_schema` (may be NULL): the deserialized TOlapProgram.Parameters batch (one row).
 *  - `device`: HIP device the program's executor is bound to.
 * On failure returns SSA_PARSE_ERROR and a malloc'd message in *err (may be NULL).
 */
int ssa_program_create(const uint8_t* program, size_t program_size, int olap_wrapper,
                       const uint32_t* col_ids, const char* const* col_names, size_t ncols,
                       const struct ArrowArray* params, const struct ArrowSchema* params_schema,
                       int device, ssa_program_t** out, char** err);
/*
 * ssa_program_create with the operations of the program's YQL kernels.  A TFunction of type YQL_KERNEL names its kernel
 * by KernelIdx, an index into the serialized MiniKQL kernel set of TProgram.Kernels (loaded by
 * ydb/library/yql/core/arrow_kernels/registry/registry.cpp:19-88); binary kernels also carry YqlOperationId, unary ones
 * (NYql::TKernelRequestBuilder::EUnaryOp, request.h:11-17: Not Size Minus Abs Just -- every pushed-down NOT is one,
 * ydb/core/kqp/query_compiler/kqp_olap_compiler.cpp:655-676, 804-812) and If do not.  This library does not read the
 * kernel set; the side that built it (TKernelRequestBuilder::AddUnaryOp / AddBinaryOp / AddIf return the index) says
 * what each index stands for: kernel_ops[i] is SSA_KERNEL_UNARY_BASE + EUnaryOp, an EBinaryOp value (request.h:19-42),
 * SSA_KERNEL_IF, or SSA_KERNEL_UNKNOWN for kernels it leaves to YqlOperationId (or that this library does not run:
 * UDFs, ScalarApply, JSON -- a program that uses one of those is refused when it is applied).  A YqlOperationId in the
 * program wins over the table.  `kernel_ops` may be NULL (nkernels 0): ssa_program_create.
 */
#define SSA_KERNEL_UNARY_BASE 0x100u
#define SSA_KERNEL_IF 0x10000u
#define SSA_KERNEL_UNKNOWN 0xFFFFFFFFu
int ssa_program_create_with_kernels(const uint8_t* program, size_t program_size, int olap_wrapper,
                                    const uint32_t* col_ids, const char* const* col_names, size_t ncols,
                                    const struct ArrowArray* params, const struct ArrowSchema* params_schema,
                                    const uint32_t* kernel_ops, size_t nkernels,
                                    int device, ssa_program_t** out, char** err);
void ssa_program_destroy(ssa_program_t* program);

/* TProgram::DebugString (ydb/core/formats/arrow/program.h:451-459); malloc'd, free with ssa_free_string. */
char* ssa_program_debug_string(const ssa_program_t* program);
/* TProgramContainer::GetSourceColumns / GetEarlyFilterColumns (tx/program/program.cpp:463-479):
 * newline-separated column names; malloc'd. */
char* ssa_program_source_columns(const ssa_program_t* program);
char* ssa_program_early_filter_columns(const ssa_program_t* program);
size_t ssa_program_step_count(const ssa_program_t* program);
/* Host-only: lowers every step against `schema` (a struct ArrowSchema; ARROW_FLAG_NULLABLE marks columns
 * that carry nulls) and returns the scan plans plus the output schema as text, or NULL and *err.  This is the
 * type-inference half of TProgramStep::Apply (which Arrow kernel DispatchBest would pick, program.cpp:51-78)
 * without touching a GPU. */
char* ssa_program_explain(const ssa_program_t* program, const struct ArrowSchema* schema, char** err);

/*
 * Whole-program seam.  Replaces NSsa::TProgram::ApplyTo(std::shared_ptr<arrow::RecordBatch>&, ExecContext*)
 * (ydb/core/formats/arrow/program.h:435-447; callers: engines/reader/plain_reader/iterator/merge.cpp:36):
 * imports the batch (honouring ArrowArray.offset of sliced columns), runs every step on the GPU and exports
 * the result.  `in` is not released; `out` / `out_schema` are filled with arrays owned by the library until
 * their release callbacks run.
 */
int ssa_program_apply(ssa_program_t* program, const struct ArrowArray* in, const struct ArrowSchema* in_schema,
                      struct ArrowArray* out, struct ArrowSchema* out_schema, char** err);

/*
 * Early-filter seam.  Replaces NSsa::TProgramStep::BuildFilter (ydb/core/formats/arrow/program.cpp:921-946,
 * caller engines/reader/plain_reader/iterator/fetching.cpp:74-86): evaluates the Assigns and Filters of step
 * `step_index` only and returns the combined filter as an Arrow-order bitmap (bit i of byte i/8, LSB first,
 * ceil(rows/8) bytes, caller-allocated) plus the number of selected rows.
 */
int ssa_program_build_filter(ssa_program_t* program, size_t step_index, const struct ArrowArray* in,
                             const struct ArrowSchema* in_schema, uint8_t* out_bits, int64_t* out_selected, char** err);

/* ---- HBM-resident batches: the same seam without the PCIe crossing ------------------------------------ */

/* Upload a RecordBatch (H2D copy of every buffer, offsets re-based). */
int ssa_batch_import(ssa_program_t* program, const struct ArrowArray* in, const struct ArrowSchema* in_schema,
                     ssa_batch_t** out, char** err);

/*
 * Column blobs as stored.  Replaces NArrow::NSerialization::TNativeSerializer::DoDeserialize
 * (ydb/core/formats/arrow/serializer/native.cpp:18-40 for a whole IPC stream written by DoSerializeFull, :60-81 for the
 * lone RecordBatch message written by DoSerializePayload and read back with the chunk's schema): the blob crosses PCIe
 * compressed and its buffers are decoded on the GPU into the columns of a resident batch.  `schema` (a struct
 * ArrowSchema, may be NULL for a whole stream) is the schema of the chunk; a stream's own schema must agree with it in
 * column count and types.  Body compression: none and LZ4_FRAME (the reference's default, serializer/native.h:54-57);
 * dictionary-encoded STRING / BINARY columns (a DictionaryBatch message per dictionary: what the dictionary transformer
 * stores, ydb/core/formats/arrow/transformer/dictionary.cpp:5-7) become dictionary columns that stay encoded (index_type
 * of ssa_device_column below); a ZSTD body, delta dictionaries, dictionaries of other value types or with null entries
 * and nested columns are SSA_UNSUPPORTED (the shim keeps the CPU deserializer for those).  Damaged metadata, buffers that do not decode to their announced length and string offsets that run past
 * the data are SSA_INVALID (the reference returns SerializationError).  ssa_batches_from_blobs decodes `nblobs` blobs
 * of one schema with one kernel launch for all their buffers (one wavefront per compressed buffer) and fills out[0 ..
 * nblobs) -- the portions of a scan; on failure no batch is returned.
 */
int ssa_batch_from_blob(ssa_program_t* program, const uint8_t* blob, size_t size, const struct ArrowSchema* schema,
                        ssa_batch_t** out, char** err);
/* Host only (needs no GPU): the metadata of a blob as text -- rows, codec, per column its type id, null count, the
 * stored lengths of its buffers and, for LZ4 frames, the announced length and block count -- or NULL and *err.  The
 * checks are those ssa_batch_from_blob makes before it touches the device. */
char* ssa_blob_describe(const uint8_t* blob, size_t size, const struct ArrowSchema* schema, char** err);
int ssa_batches_from_blobs(ssa_program_t* program, const uint8_t* const* blobs, const size_t* sizes, size_t nblobs,
                           const struct ArrowSchema* schema, ssa_batch_t** out, char** err);

/* Column already in device memory (not copied, not owned; must outlive the batch).  The library reads it on its own
 * HIP stream: whatever produced the column (a copy, a kernel on another stream) must have completed before the batch is
 * used. */
struct ssa_device_column {
    const char* name;
    int32_t type;          /* enum ssa_type */
    int32_t index_type;    /* 0: a plain column.  An integer enum ssa_type: a dictionary-encoded STRING / BINARY column
                            * (arrow::DictionaryArray, which the reference decodes when it loads a chunk,
                            * ydb/core/formats/arrow/dictionary/conversion.h:13-15): `values` holds one code of this
                            * type per row, `chars` / `dict_offsets` / `dict_size` the dictionary.  Programs give the
                            * results of the decoded column; predicates and GROUP BY keys are worked on the codes.  A
                            * code outside [0, dict_size) reads as a null. */
    const void* values;    /* fixed width: values; BOOL: bitmap; STRING/BINARY: int32 offsets (rows + 1); dictionary: codes */
    const uint8_t* validity; /* bitmap, NULL = no nulls */
    const uint8_t* chars;  /* STRING/BINARY bytes (of the dictionary's entries when index_type != 0); the allocation must
                            * be readable for 16 bytes past chars_size (the scans read strings in 16-byte pieces) */
    int64_t chars_size;
    int64_t null_count;    /* -1 = unknown */
    const int32_t* dict_offsets; /* dictionary: int32 offsets of the entries into chars (dict_size + 1), else NULL */
    int64_t dict_size;     /* dictionary: number of entries */
};
int ssa_batch_from_device(const struct ssa_device_column* cols, size_t ncols, int64_t rows, ssa_batch_t** out, char** err);
/* Process-wide counters of the work on dictionary-encoded columns (no counterpart in the reference, which decodes
 * every such column when it loads it, ydb/core/formats/arrow/transformer/dictionary.cpp:9-11): program steps and
 * filters that ran over the codes, and whole columns that had to be decoded (a gather through the codes) because a step
 * used them as values. */
void ssa_dictionary_stats(int64_t* steps_on_codes, int64_t* columns_decoded);

/*
 * Operator-level seam.  Replaces the Arrow registry function "ch.group_by" (CH::ArrowGroupBy,
 * ydb/library/arrow_clickhouse/AggregateFunctions/AggregateFunctionWrapper.h:100-255, registered at
 * ydb/core/formats/arrow/custom_registry.cpp:65-82, called by TProgramStep::ApplyAggregates,
 * ydb/core/formats/arrow/program.cpp:722-750) for callers that hold a CH::GroupByOptions rather than a program:
 * `assigns` is GroupByOptions::assigns (IAggregateFunction.h:498-511) -- aggregates and keys in the order the result is
 * to have; `function` is CH::AggFunctionId (IAggregateFunction.h:478-485): 0 marks a key named by `result_column`, 1 any,
 * 2 count, 3 min, 4 max, 5 sum; `argument` is the aggregate's one argument column (NULL: count of rows).  Anything else
 * (avg and the ids the reference leaves commented out) is "Unexpected agregate function in GROUP BY".  The handle is a
 * program of one aggregation like any other: apply it with ssa_program_apply / ssa_program_apply_device /
 * ssa_program_apply_many to the datums' batch (GroupByOptions::schema is the batch's own; has_nullable_key is read off
 * the key columns, as ApplyAggregates does), destroy it with ssa_program_destroy.  Kernels built for it stay with the
 * handle, so keep it for as long as the same aggregation comes back.
 */
struct ssa_group_by_assign {
    int32_t function;
    const char* result_column;
    const char* argument;
};
int ssa_group_by_create(const struct ssa_group_by_assign* assigns, size_t n_assigns, int device, ssa_program_t** out, char** err);

/* TProgram::ApplyTo on an HBM-resident batch; the result stays in HBM. */
int ssa_program_apply_device(ssa_program_t* program, const ssa_batch_t* in, ssa_batch_t** out, char** err);
/* TProgram::ApplyTo on each of `n` HBM-resident batches -- the slices of a table, which the reference runs one by one
 * (ydb/core/formats/arrow/program.h:423-433) -- with one result per slice in out[0 .. n), not merged.  Slices of one
 * schema whose step is an aggregation (GROUP BY over fixed-width keys or string keys of one length, or no keys) are
 * scanned by one kernel launch with one host synchronisation for all of them; anything else runs slice by slice, with
 * the same results.  On failure no batch is returned. */
int ssa_program_apply_many(ssa_program_t* program, const ssa_batch_t* const* in, size_t n, ssa_batch_t** out, char** err);
/*
 * Result batches between the GPUs of a sharded scan.  The reference sends every shard's partial result to the compute
 * node that merges them (ydb/core/tx/columnshard/columnshard__scan.cpp:300-340, TEvScanData); with one process per GPU
 * the partial tables stay in device memory and travel as flat buffers through a collective of the host's choice
 * (RCCL all_gather / all_to_all):
 *  - ssa_batch_pack_layout: the description of the batch as a flat buffer -- `header` (host bytes; NULL to ask for the
 *    sizes only) and the size of its data (string columns: the bytes in use, read from the first and last offset);
 *  - ssa_batch_pack_data: the column buffers into `device_dst` (device memory of at least data_bytes), one launch;
 *  - ssa_batch_unpack: a batch whose columns are views of `device_data` as described by `header`; the caller keeps the
 *    memory alive, and unchanged, while the batch or anything derived from it without a copy is in use;
 *  - ssa_batch_concat: the rows of `n` batches of one schema (names and types; nulls may be present in some only) as one
 *    new batch, part after part, one launch for all buffers;
 *  - ssa_batch_slice: rows [row0, row0 + rows) as a view; ssa_batch_rename: the same columns under other names (the
 *    merge program reads the partial aggregates under names of its own); ssa_batch_column_name: to be freed with
 *    ssa_free_string.
 */
int ssa_batch_pack_layout(ssa_program_t* program, const ssa_batch_t* batch, uint8_t* header, size_t header_capacity, size_t* header_bytes,
                          size_t* data_bytes, char** err);
int ssa_batch_pack_data(ssa_program_t* program, const ssa_batch_t* batch, void* device_dst, size_t capacity, char** err);
int ssa_batch_unpack(const uint8_t* header, size_t header_bytes, void* device_data, size_t data_bytes, ssa_batch_t** out, char** err);
int ssa_batch_concat(ssa_program_t* program, const ssa_batch_t* const* parts, size_t n, ssa_batch_t** out, char** err);
int ssa_batch_slice(const ssa_batch_t* batch, int64_t row0, int64_t rows, ssa_batch_t** out, char** err);
int ssa_batch_rename(const ssa_batch_t* batch, const char* const* names, size_t n, ssa_batch_t** out, char** err);
char* ssa_batch_column_name(const ssa_batch_t* batch, size_t i);

/* Process-wide counters: program steps ssa_program_apply_many ran over all their slices with one launch, and steps it
 * ran slice by slice. */
void ssa_apply_many_stats(int64_t* steps_in_one_launch, int64_t* steps_slice_by_slice);
/* Process-wide counter: GROUP BY scans that applied the rows missing the workgroups' own tables region by region of the
 * table in HBM (groupings of a million groups and more take this way; SSA_KEYED_PARTITION=0 / 1 forbids / forces it). */
void ssa_keyed_stats(int64_t* scans_applied_by_region);
/* Process-wide counter: of those, the scans whose records were first sorted without counting passes (every region's place
 * in the sorted arrays gets room to spare), found a place too small for its region and were sorted again with counts --
 * keys that crowd into few regions of the table do that; the step sorts with counts from then on. */
void ssa_keyed_sort_stats(int64_t* sorts_done_again_with_counts);
/* Process-wide counter: GROUP BY scans run by the kernel that deals the rows that passed the step's filters out evenly
 * over the lanes of a wave and aggregates them one by one (steps behind a selective filter whose keys and aggregate
 * arguments are plain columns without nulls; timed against the other ways on the step's first large scans,
 * SSA_KEYED_HOT=2 forces it). */
void ssa_keyed_compact_stats(int64_t* scans_with_compacted_tiles);
/* Download a batch into Arrow C Data structures. */
int ssa_batch_export(ssa_program_t* program, const ssa_batch_t* batch, struct ArrowArray* out,
                     struct ArrowSchema* out_schema, char** err);
int64_t ssa_batch_rows(const ssa_batch_t* batch);
size_t ssa_batch_columns(const ssa_batch_t* batch);
void ssa_batch_destroy(ssa_batch_t* batch);

/* ---- compute sharding of a result batch --------------------------------------------------------------------- */

/* NArrow::NHash::TXX64::Execute (ydb/core/formats/arrow/hash/calcer.cpp:52-70): out_hashes[row] (host array of
 * ssa_batch_rows(batch) values) = XXH64 with `seed` over the concatenated bytes of the row's non-null key fields, in the
 * order of key_columns (integers, floats, timestamps: their value bytes; booleans: one byte; strings: their bytes;
 * decimals: nothing).  A key column that is not in the batch is SSA_INVALID (ENoColumnPolicy::Verify). */
int ssa_sharding_hash(ssa_program_t* program, const ssa_batch_t* batch, const char* const* key_columns, size_t n_keys,
                      uint64_t seed, uint64_t* out_hashes, char** err);

/* THashConstructor::BuildHashUI64 + TShardingSplitIndex::Apply (ydb/core/formats/arrow/permutations.cpp:187-207 and
 * :259-287, permutations.h:84-125; call site engines/reader/plain_reader/iterator/merge.cpp:38-45): the hash of a row
 * is the key column itself when there is exactly one key of type (U)Int32/(U)Int64, else XXH64 with seed 34323543;
 * shard = |hash| % shards_count; the per-shard row lists are balanced as the reference does (records move from the
 * longest list to the shortest while one is above and one below rows / shards_count + 1).  *out is the input with its
 * rows reordered shard after shard, rows of a shard in ascending input order (what Take(input, BuildPermutation())
 * returns), out_shard_rows[s] of them in shard s.  A batch the reference leaves unsharded (one shard, no key columns,
 * a single missing key column) comes back unchanged with every row counted in shard 0.  Lists of equal length are
 * ordered by shard number where the reference's std::sort leaves the order open.  Up to 512 shards. */
int ssa_sharding_split(ssa_program_t* program, const ssa_batch_t* batch, const char* const* key_columns, size_t n_keys,
                       uint32_t shards_count, ssa_batch_t** out, uint32_t* out_shard_rows, char** err);

/* The same split without the balancing of the lists: part = |hash| % parts_count and nothing else, so that rows with equal
 * keys are in the same part whichever batch they come from.  No counterpart in the reference (its balanced lists are
 * what a KQP compute task reads, and equal keys may end in different ones); this is the partitioning step of the
 * exchange between GPUs, where every rank must send a group's partial aggregates to the one rank that merges it. */
int ssa_batch_partition_by_key(ssa_program_t* program, const ssa_batch_t* batch, const char* const* key_columns, size_t n_keys,
                               uint32_t parts_count, ssa_batch_t** out, uint32_t* out_part_rows, char** err);

/* ---- TColumnFilter: filters kept in HBM and their algebra ---------------------------------------------------------- */

/* NArrow::TColumnFilter (ydb/core/formats/arrow/arrow_filter.h:20-211) with its bits in device memory.  The reference
 * keeps run lengths of alternating values; the handle keeps a bitmap and gives the run lengths back on request.  A
 * filter "without a size" (Filter.empty(): BuildAllowFilter / BuildDenyFilter, or one made from no rows) stays that. */
typedef struct ssa_filter ssa_filter_t;

/* ECompareType (arrow_filter.h:13-18) */
enum ssa_compare_type { SSA_CMP_LESS = 1, SSA_CMP_LESS_OR_EQUAL = 2, SSA_CMP_GREATER = 3, SSA_CMP_GREATER_OR_EQUAL = 4 };

/* TColumnFilter::BuildAllowFilter / BuildDenyFilter (arrow_filter.h:188-194).  Needs no GPU. */
int ssa_filter_simple(int allow, ssa_filter_t** out);
/* TColumnFilter(std::vector<bool>&&) (arrow_filter.h:132-155) with the values given as runs: run j has runs[j] rows of
 * value first_value ^ (j & 1).  Runs of length 0 are skipped and equal neighbours joined, as TColumnFilter::Add does
 * (arrow_filter.cpp:239-250); no rows at all give the allow filter without a size. */
int ssa_filter_from_runs(ssa_program_t* program, int first_value, const uint32_t* runs, size_t n_runs, ssa_filter_t** out, char** err);
/* TProgramStep::BuildFilter (ydb/core/formats/arrow/program.cpp:921-946) on an HBM-resident batch, the result staying in
 * HBM.  *out = NULL when the step has no Filter command (the reference returns nullptr). */
int ssa_filter_from_program(ssa_program_t* program, size_t step_index, const ssa_batch_t* batch, ssa_filter_t** out, char** err);
/* No counterpart in the reference.  For callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc): ssa_filter_from_program with the result in caller-provided
 * memory.  `out_bits` is DEVICE memory of ceil(rows / 64) 64-bit words and receives the filter as an Arrow-order
 * bitmap (bit i of word i / 64 = row i passes, zero past the last row); *out_selected the number of rows kept.  The
 * scan's per-tile counts go to a static buffer of 256 KiB + 256 bytes that every GPU the library runs on holds for
 * the life of the process; the generic scan kernel runs (none is built for the program); no hipMalloc / hipFree is
 * called.  At most 67 108 864 (2^26) rows: more is SSA_INVALID.  A step without a Filter command is SSA_INVALID.
 * Calls on one GPU share that GPU's buffer: calls that overlap on one GPU -- from programs of their own too -- race;
 * serialise them. */
int ssa_filter_from_program_noalloc(ssa_program_t* program, size_t step_index, const ssa_batch_t* batch, uint64_t* out_bits,
                                    int64_t* out_selected, char** err);
/* TColumnFilter::MakePredicateFilter(batch, border, compareType) (arrow_filter.cpp:118-136, 259-311; caller
 * engines/predicate/container.h:74): `border` is a one-row RecordBatch (host memory, Arrow C Data Interface) whose
 * columns are the leading key columns; rows are compared with it key after key (values only: validity is not looked
 * at, strings by memcmp then length) and kept according to `compare_type` (enum ssa_compare_type).  A border column
 * that is missing from the batch or differs in type is SSA_INVALID (the reference aborts). */
int ssa_filter_predicate(ssa_program_t* program, const ssa_batch_t* batch, const struct ArrowArray* border,
                         const struct ArrowSchema* border_schema, int compare_type, ssa_filter_t** out, char** err);
/* TColumnFilter::And / Or (arrow_filter.cpp:402-513).  Two sized filters must have the same size (SSA_INVALID
 * otherwise; the reference aborts). */
int ssa_filter_and(ssa_program_t* program, const ssa_filter_t* a, const ssa_filter_t* b, ssa_filter_t** out, char** err);
int ssa_filter_or(ssa_program_t* program, const ssa_filter_t* a, const ssa_filter_t* b, ssa_filter_t** out, char** err);
/* TColumnFilter::CombineSequentialAnd (arrow_filter.cpp:515-575; caller reader/plain_reader/iterator/fetched_data.h:70-80):
 * `ext` has one row for every row `self` keeps and decides which of those stay. */
int ssa_filter_combine_sequential_and(ssa_program_t* program, const ssa_filter_t* self, const ssa_filter_t* ext,
                                      ssa_filter_t** out, char** err);
/* State as the reference holds it (arrow_filter.h:22-25): *is_empty = Filter.empty(), *default_value =
 * DefaultFilterValue, *count = Count (Size()).  Needs no GPU. */
void ssa_filter_info(const ssa_filter_t* filter, int* is_empty, int* default_value, int64_t* count);
/* The `Filter` vector and LastValue: call with runs = NULL to learn *n_runs, then with an array of that many values. */
int ssa_filter_runs(ssa_program_t* program, const ssa_filter_t* filter, uint32_t* runs, size_t* n_runs, int* last_value, char** err);
/* TColumnFilter::BuildSimpleFilter (arrow_filter.cpp:383-400) as an Arrow-order bitmap: ceil(count / 8) bytes. */
int ssa_filter_bits(ssa_program_t* program, const ssa_filter_t* filter, uint8_t* out_bits, char** err);
/* TColumnFilter::GetFilteredCount (arrow_filter.cpp:586-608); *out = -1 where the reference returns an empty optional
 * (the allow filter without a size). */
int ssa_filter_filtered_count(ssa_program_t* program, const ssa_filter_t* filter, int64_t* out, char** err);
/* TColumnFilter::Apply(std::shared_ptr<arrow::RecordBatch>&) (arrow_filter.cpp:313-348): the rows the filter keeps, in
 * order, every column compacted on the GPU. */
int ssa_filter_apply(ssa_program_t* program, const ssa_filter_t* filter, const ssa_batch_t* batch, ssa_batch_t** out, char** err);
void ssa_filter_destroy(ssa_filter_t* filter);

/*
 * NArrow::NMerger::TMergePartialStream (ydb/core/formats/arrow/reader/merger.h:41-95): AddSource for each of the
 * `n_sources` batches (every one sorted ascending by `key_columns`; `filters`: NULL, or per source a filter or NULL)
 * followed by DrainAll (merger.cpp:145-186).  The result is sorted by the key columns -- descending when `reverse`, as
 * the reference reads its sources from their ends then -- and holds every key once: the row with the greatest
 * `version_columns` among the rows of that key, and no row at all when that row is denied by its source's filter.
 * Rows compare as in ydb/core/formats/arrow/switch/compare.h:82-119 (validity is not looked at, strings by memcmp then
 * length).  Sources without rows and sources whose filter denies everything take no part (merger.h:80-85).  Among rows of
 * one key and one version -- an order the reference's heap leaves open -- the row of the earlier source is taken.
 * All columns of the sources are carried (they must agree in names and types).
 */
int ssa_merge_sorted(ssa_program_t* program, const ssa_batch_t* const* sources, const ssa_filter_t* const* filters, size_t n_sources,
                     const char* const* key_columns, size_t n_keys, const char* const* version_columns, size_t n_versions, int reverse,
                     ssa_batch_t** out, char** err);

/* ---- measurement ------------------------------------------------------------------------------------------ */

/* When enabled, every fused scan kernel launch of this program is bracketed by HIP events on the program's
 * stream.  ssa_profile_read returns and clears: accumulated kernel milliseconds, launches, and algorithmic
 * bytes (each referenced source column counted once per launch). */
void ssa_profile_enable(ssa_program_t* program, int enabled);
void ssa_profile_read(ssa_program_t* program, double* scan_ms, int64_t* launches, int64_t* algorithmic_bytes);

/* Scan kernels built at run time for one step program (no counterpart in the reference's C++ executor; the
 * reference compiles MiniKQL graphs with LLVM in ydb/library/yql/minikql/codegen).  A step program whose scans add
 * up to 2^25 rows gets its scan kernel compiled for it through hiprtc: the generic kernel's source with the
 * program's tables as constants.  Environment SSA_JIT=0 turns this off, SSA_JIT=1 builds at the first scan.  The
 * counters are process-wide: kernels built, host milliseconds spent building, launches that ran such a kernel. */
void ssa_specialization_stats(int64_t* kernels_built, double* build_ms, int64_t* launches);

/* ---- synthetic TPC-H lineitem generated in HBM (benchmark / parity input) ----------------------------------- */

#ifdef __cplusplus
}
#endif

#endif /* YDB_SSA_HIP_H */
