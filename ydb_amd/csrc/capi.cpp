// extern "C" boundary: see include/ydb_ssa_hip.h for the contract of every entry point.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>

#include "../../include/ydb_ssa_hip.h"
#include "exec.h"
#include "ipc.h"
#include "jit.h"
#include "program.h"
#include "vm_ops.h"

using namespace ssa;

struct ssa_program {
    TProgramContainer Container;
    std::unique_ptr<Executor> Exec;
    std::mutex Lock;  // one HIP stream per program: concurrent applies on one handle are serialised
    int Device = 0;
};

struct ssa_batch {
    BatchPtr Batch;
};

struct ssa_filter {
    FilterPtr Filter;
};

namespace {

char* DupString(const std::string& s) {
    char* p = static_cast<char*>(std::malloc(s.size() + 1));
    if (p) std::memcpy(p, s.c_str(), s.size() + 1);
    return p;
}

int Fail(const Status& st, char** err) {
    if (err) *err = DupString(st.Msg);
    switch (st.Code) {
        case ST_OK: return SSA_OK;
        case ST_INVALID: return SSA_INVALID;
        case ST_PARSE_ERROR: return SSA_PARSE_ERROR;
        case ST_UNSUPPORTED: return SSA_UNSUPPORTED;
        default: return SSA_HIP_ERROR;
    }
}

// The columns of a one-row batch as named scalars, in schema order.
Status ReadRowScalars(const ArrowArray* arr, const ArrowSchema* schema, std::vector<std::pair<std::string, TScalar>>& out) {
    if (!arr || !schema) return Status::OK();
    if (!schema->format || std::string(schema->format) != "+s") return Status::Invalid("expected a struct array (an exported one-row RecordBatch)");
    for (int64_t i = 0; i < arr->n_children; ++i) {
        const ArrowArray* a = arr->children[i];
        const ArrowSchema* s = schema->children[i];
        if (a->length != 1) continue;  // "Incorrect values count in parameter array" -> parameter unusable
        const std::string fmt = s->format ? s->format : "";
        TScalar sc;
        const int64_t off = a->offset;
        const uint8_t* validity = a->n_buffers > 0 ? static_cast<const uint8_t*>(a->buffers[0]) : nullptr;
        sc.Valid = !(validity && a->null_count != 0 && !((validity[off >> 3] >> (off & 7)) & 1));
        const void* v = a->n_buffers > 1 ? a->buffers[1] : nullptr;
        auto num = [&](int type, uint64_t bits) {
            sc.Type = type;
            sc.Bits = bits;
        };
        if (fmt == "b") num(T_BOOL, (static_cast<const uint8_t*>(v)[off >> 3] >> (off & 7)) & 1);
        else if (fmt == "c") num(T_I8, uint64_t(int64_t(static_cast<const int8_t*>(v)[off])));
        else if (fmt == "C") num(T_U8, static_cast<const uint8_t*>(v)[off]);
        else if (fmt == "s") num(T_I16, uint64_t(int64_t(static_cast<const int16_t*>(v)[off])));
        else if (fmt == "S") num(T_U16, static_cast<const uint16_t*>(v)[off]);
        else if (fmt == "i") num(T_I32, uint64_t(int64_t(static_cast<const int32_t*>(v)[off])));
        else if (fmt == "I") num(T_U32, static_cast<const uint32_t*>(v)[off]);
        else if (fmt == "l") num(T_I64, uint64_t(static_cast<const int64_t*>(v)[off]));
        else if (fmt == "L") num(T_U64, static_cast<const uint64_t*>(v)[off]);
        else if (fmt == "f") num(T_F32, F64ToBits(double(static_cast<const float*>(v)[off])));
        else if (fmt == "g") num(T_F64, F64ToBits(static_cast<const double*>(v)[off]));
        else if (fmt.rfind("tsu:", 0) == 0) num(T_TS, uint64_t(static_cast<const int64_t*>(v)[off]));
        else if (fmt == "tDu") num(T_DUR, uint64_t(static_cast<const int64_t*>(v)[off]));
        else if (fmt == "u" || fmt == "z") {
            sc.Type = fmt == "u" ? T_STRING : T_BINARY;
            const int32_t* offs = static_cast<const int32_t*>(a->buffers[1]);
            const char* chars = static_cast<const char*>(a->buffers[2]);
            sc.Bytes.assign(chars + offs[off], size_t(offs[off + 1] - offs[off]));
        } else {
            continue;
        }
        out.emplace_back(s->name ? s->name : "", sc);
    }
    return Status::OK();
}

// First row of a one-row parameters batch -> named scalars (tx/program/program.cpp:338-356).
Status ReadParameters(const ArrowArray* arr, const ArrowSchema* schema, TParameters& out) {
    std::vector<std::pair<std::string, TScalar>> row;
    Status st = ReadRowScalars(arr, schema, row);
    if (!st.ok()) return st;
    for (auto& kv : row) out[kv.first] = kv.second;
    return Status::OK();
}

// Parsing needs no GPU; executing does.  There is deliberately no CPU fallback behind this boundary.
Status EnsureExecutor(ssa_program* p) {
    if (p->Exec) return Status::OK();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        return Status::Hip("no HIP device: the GPU executor cannot run here (this library has no CPU fallback)");
    }
    if (p->Device < 0 || p->Device >= ndev) return Status::Invalid("bad device index");
    p->Exec = std::make_unique<Executor>(p->Device);
    return Status::OK();
}

char* JoinNames(const std::set<std::string>& names) {
    std::string s;
    for (auto& n : names) {
        if (!s.empty()) s += "\n";
        s += n;
    }
    return DupString(s);
}

}  // namespace

extern "C" {

int ssa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -SSA_HIP_ERROR;
    return n;
}

const char* ssa_version(void) { return "ydb-ssa-hip 0.1 (gfx950)"; }

void ssa_free_string(char* s) { std::free(s); }

int ssa_program_create(const uint8_t* program, size_t program_size, int olap_wrapper, const uint32_t* col_ids,
                       const char* const* col_names, size_t ncols, const ArrowArray* params,
                       const ArrowSchema* params_schema, int device, ssa_program_t** out, char** err) {
    return ssa_program_create_with_kernels(program, program_size, olap_wrapper, col_ids, col_names, ncols, params, params_schema, nullptr, 0,
                                           device, out, err);
}

int ssa_program_create_with_kernels(const uint8_t* program, size_t program_size, int olap_wrapper, const uint32_t* col_ids,
                                    const char* const* col_names, size_t ncols, const ArrowArray* params,
                                    const ArrowSchema* params_schema, const uint32_t* kernel_ops, size_t nkernels, int device,
                                    ssa_program_t** out, char** err) {
    if (err) *err = nullptr;
    if (!out) return SSA_INVALID;
    *out = nullptr;
    TMapColumnResolver resolver;
    for (size_t i = 0; i < ncols; ++i) resolver.Add(col_ids[i], col_names[i] ? col_names[i] : "");
    TParameters parameters;
    Status st = ReadParameters(params, params_schema, parameters);
    if (!st.ok()) return Fail(st, err);
    auto p = std::make_unique<ssa_program>();
    std::string error;
    std::vector<uint32_t> kernelOps;
    if (kernel_ops != nullptr) kernelOps.assign(kernel_ops, kernel_ops + nkernels);
    for (uint32_t op : kernelOps) {
        const bool known = op <= uint32_t(EYqlOp::Coalesce) || (op >= kYqlUnaryBase && op <= uint32_t(EYqlOp::Just)) || op == uint32_t(EYqlOp::If) ||
                           op == kYqlKernelUnknown;
        if (!known) return Fail(Status{ST_INVALID, "kernel_ops: " + std::to_string(op) + " names no kernel operation"}, err);
    }
    if (!p->Container.Init(resolver, program, program_size, (olap_wrapper & 1) != 0, parameters, error, (olap_wrapper & SSA_PROGRAM_DIRECT_OPERATIONS) != 0,
                           kernel_ops != nullptr ? &kernelOps : nullptr)) {
        return Fail(Status{ST_PARSE_ERROR, error}, err);
    }
    p->Device = device;  // the executor (HIP stream, device buffers) is created at first use
    *out = p.release();
    return SSA_OK;
}

// CH::ArrowGroupBy (AggregateFunctionWrapper.h:100-255) as a program of one step: the aggregates and the keys of the
// options, and -- when the result is to have another order than a GroupBy command's (aggregates, then keys) -- a
// projection behind it.
int ssa_group_by_create(const ssa_group_by_assign* assigns, size_t n_assigns, int device, ssa_program_t** out, char** err) {
    if (err) *err = nullptr;
    if (!out) return SSA_INVALID;
    *out = nullptr;
    if (!assigns || n_assigns == 0) return Fail(Status{ST_INVALID, "GROUP BY without options"}, err);
    auto program = std::make_shared<TProgram>();
    auto step = std::make_shared<TProgramStep>();
    auto source = [&](const std::string& name) {
        for (auto& kv : program->SourceColumns) {
            if (kv.second.ColumnName == name) return kv.second;
        }
        const uint32_t id = uint32_t(program->SourceColumns.size()) + 1u;
        program->SourceColumns.emplace_back(id, TColumnInfo::Original(id, name));
        return program->SourceColumns.back().second;
    };
    std::vector<TColumnInfo> order;  // the result's columns as the caller listed them
    std::set<std::string> results;
    size_t naggs = 0;
    for (size_t i = 0; i < n_assigns; ++i) {
        const ssa_group_by_assign& a = assigns[i];
        if (!a.result_column || !a.result_column[0]) return Fail(Status{ST_INVALID, "GROUP BY: assign without a result column"}, err);
        if (!results.insert(a.result_column).second) return Fail(Status{ST_INVALID, std::string("GROUP BY: column '") + a.result_column + "' assigned twice"}, err);
        if (a.function == 0) {
            step->GroupByKeys.push_back(source(a.result_column));
            order.push_back(step->GroupByKeys.back());
            continue;
        }
        if (a.function < int32_t(EAggregate::Some) || a.function > int32_t(EAggregate::Sum)) {
            return Fail(Status{ST_INVALID, "Unexpected agregate function in GROUP BY"}, err);
        }
        TAggregateAssign g;
        g.Column = TColumnInfo::Generated(10000u + uint32_t(i), a.result_column);
        g.Operation = EAggregate(a.function);
        if (a.argument && a.argument[0]) g.Arguments.push_back(source(a.argument));
        else if (g.Operation != EAggregate::Count) return Fail(Status{ST_INVALID, "Unexpected aggregate function argument in GROUP BY"}, err);
        step->GroupBy.push_back(g);
        order.push_back(g.Column);
        ++naggs;
    }
    if (naggs == 0) return Fail(Status{ST_UNSUPPORTED, "GROUP BY without aggregates (the keys' distinct values)"}, err);
    program->Steps.push_back(step);
    bool natural = true;  // aggregates first, keys behind them: what the step produces
    for (size_t i = 0; i < n_assigns; ++i) natural = natural && ((assigns[i].function != 0) == (i < naggs));
    if (!natural) {
        auto projection = std::make_shared<TProgramStep>();
        projection->Projection = order;
        program->Steps.push_back(projection);
    }
    if (program->SourceColumns.empty()) {  // count of rows and nothing else: any column of the batch will do
        return Fail(Status{ST_UNSUPPORTED, "GROUP BY that reads no column"}, err);
    }
    auto p = std::make_unique<ssa_program>();
    p->Container.Adopt(program);
    p->Device = device;
    *out = p.release();
    return SSA_OK;
}

void ssa_program_destroy(ssa_program_t* program) { delete program; }

char* ssa_program_debug_string(const ssa_program_t* program) {
    return DupString(program->Container.GetProgram()->DebugString());
}

char* ssa_program_source_columns(const ssa_program_t* program) {
    return JoinNames(program->Container.GetProgram()->GetProcessingColumns());
}

char* ssa_program_early_filter_columns(const ssa_program_t* program) {
    return JoinNames(program->Container.GetProgram()->GetEarlyFilterColumns());
}

size_t ssa_program_step_count(const ssa_program_t* program) { return program->Container.GetProgram()->Steps.size(); }

char* ssa_program_explain(const ssa_program_t* program, const ArrowSchema* schema, char** err) {
    if (err) *err = nullptr;
    std::vector<FieldDesc> fields;
    Status st = SchemaFields(schema, fields);
    if (!st.ok()) {
        Fail(st, err);
        return nullptr;
    }
    std::string text;
    for (auto& step : program->Container.GetProgram()->Steps) {
        StepPlan plan;
        st = CompileStep(*step, fields, CM_FULL, plan);
        if (!st.ok()) {
            Fail(st, err);
            return nullptr;
        }
        text += plan.DebugString() + "\n";
        std::vector<FieldDesc> next;
        for (auto& oc : plan.Outs) {
            FieldDesc f;
            f.Name = oc.Name;
            f.Type = oc.Type;
            f.Nullable = oc.Nullable;
            next.push_back(f);
        }
        fields = next;
    }
    text += "schema:";
    for (auto& f : fields) text += " " + f.Name + ":" + std::to_string(f.Type);
    return DupString(text);
}

int ssa_program_apply(ssa_program_t* program, const ArrowArray* in, const ArrowSchema* in_schema, ArrowArray* out,
                      ArrowSchema* out_schema, char** err) {
    if (err) *err = nullptr;
    std::lock_guard<std::mutex> g(program->Lock);
    {
        Status est = EnsureExecutor(program);
        if (!est.ok()) return Fail(est, err);
    }
    BatchPtr batch, result;
    Status st = program->Exec->ImportBatch(in, in_schema, batch);
    if (!st.ok()) return Fail(st, err);
    st = program->Exec->ApplyProgram(*program->Container.GetProgram(), batch, result);
    if (!st.ok()) return Fail(st, err);
    st = program->Exec->ExportBatch(*result, out, out_schema);
    if (!st.ok()) return Fail(st, err);
    return SSA_OK;
}

int ssa_program_build_filter(ssa_program_t* program, size_t step_index, const ArrowArray* in, const ArrowSchema* in_schema,
                             uint8_t* out_bits, int64_t* out_selected, char** err) {
    if (err) *err = nullptr;
    std::lock_guard<std::mutex> g(program->Lock);
    {
        Status est = EnsureExecutor(program);
        if (!est.ok()) return Fail(est, err);
    }
    auto& steps = program->Container.GetProgram()->Steps;
    if (step_index >= steps.size()) return Fail(Status::Invalid("no such step"), err);
    BatchPtr batch;
    Status st = program->Exec->ImportBatch(in, in_schema, batch);
    if (!st.ok()) return Fail(st, err);
    if (steps[step_index]->Filters.empty()) {  // BuildFilter returns nullptr: everything passes
        std::memset(out_bits, 0xFF, size_t((batch->Rows + 7) / 8));
        if (batch->Rows & 7) out_bits[batch->Rows / 8] = uint8_t((1u << (batch->Rows & 7)) - 1);
        if (out_selected) *out_selected = batch->Rows;
        return SSA_OK;
    }
    std::vector<uint8_t> bits;
    int64_t selected = 0;
    st = program->Exec->BuildFilter(*steps[step_index], batch, bits, selected);
    if (!st.ok()) return Fail(st, err);
    std::memcpy(out_bits, bits.data(), bits.size());
    if (out_selected) *out_selected = selected;
    return SSA_OK;
}

int ssa_batch_import(ssa_program_t* program, const ArrowArray* in, const ArrowSchema* in_schema, ssa_batch_t** out, char** err) {
    if (err) *err = nullptr;
    std::lock_guard<std::mutex> g(program->Lock);
    {
        Status est = EnsureExecutor(program);
        if (!est.ok()) return Fail(est, err);
    }
    BatchPtr batch;
    Status st = program->Exec->ImportBatch(in, in_schema, batch);
    if (!st.ok()) return Fail(st, err);
    *out = new ssa_batch{batch};
    return SSA_OK;
}

int ssa_batches_from_blobs(ssa_program_t* program, const uint8_t* const* blobs, const size_t* sizes, size_t nblobs,
                           const ArrowSchema* schema, ssa_batch_t** out, char** err) {
    if (err) *err = nullptr;
    if (!program || (nblobs && (!blobs || !sizes || !out))) return Fail(Status::Invalid("null argument"), err);
    std::lock_guard<std::mutex> g(program->Lock);
    {
        Status est = EnsureExecutor(program);
        if (!est.ok()) return Fail(est, err);
    }
    std::vector<FieldDesc> fields;
    if (schema) {
        Status st = SchemaFields(schema, fields);
        if (!st.ok()) return Fail(st, err);
    }
    std::vector<std::pair<const uint8_t*, size_t>> list;
    for (size_t i = 0; i < nblobs; ++i) list.emplace_back(blobs[i], sizes[i]);
    std::vector<BatchPtr> batches;
    Status st = program->Exec->ImportBlobs(list, fields, batches);
    if (!st.ok()) return Fail(st, err);
    for (size_t i = 0; i < nblobs; ++i) out[i] = new ssa_batch{batches[i]};
    return SSA_OK;
}

char* ssa_blob_describe(const uint8_t* blob, size_t size, const ArrowSchema* schema, char** err) {
    if (err) *err = nullptr;
    std::vector<FieldDesc> fields;
    if (schema) {
        Status st = SchemaFields(schema, fields);
        if (!st.ok()) {
            Fail(st, err);
            return nullptr;
        }
    }
    IpcBatch b;
    Status st = ParseIpcBlob(blob, size, fields, b);
    if (!st.ok()) {
        Fail(st, err);
        return nullptr;
    }
    std::string text = "rows=" + std::to_string(b.Rows) + " codec=" + (b.Codec == IPC_RAW ? "none" : b.Codec == IPC_LZ4_FRAME ? "lz4_frame" : "zstd") +
                       " body=" + std::to_string(b.BodyLength) + "\n";
    for (const IpcColumn& c : b.Columns) {
        text += c.Field.Name + ":" + std::to_string(c.Field.Type) + " nulls=" + std::to_string(c.NullCount) + " buffers=" +
                std::to_string(c.Validity.Length) + "," + std::to_string(c.Values.Length) + "," + std::to_string(c.Data.Length);
        if (c.DictId >= 0) {
            for (const IpcDictionary& d : b.Dicts) {
                if (d.Id == c.DictId) {
                    text += " dictionary(id " + std::to_string(d.Id) + ", " + std::to_string(d.Rows) + " entries, codes of type " + std::to_string(c.IndexType) + ")";
                }
            }
        }
        if (b.Codec == IPC_LZ4_FRAME) {
            for (const IpcBuffer* buf : {&c.Validity, &c.Values, &c.Data}) {
                if (buf->Length <= 8) continue;
                int64_t ulen;
                std::memcpy(&ulen, blob + b.BodyOffset + buf->Offset, 8);
                if (ulen < 0) continue;
                std::vector<Lz4BlockRef> blocks;
                uint32_t blockMax = 0;
                bool linked = false;
                st = ParseLz4Frame(blob, size, b.BodyOffset + size_t(buf->Offset) + 8, size_t(buf->Length - 8), blocks, blockMax, linked);
                if (!st.ok()) {
                    Fail(st, err);
                    return nullptr;
                }
                text += " frame(" + std::to_string(ulen) + " bytes, " + std::to_string(blocks.size()) + " blocks of " + std::to_string(blockMax) +
                        (linked ? ", linked)" : ", independent)");
            }
        }
        text += "\n";
    }
    return DupString(text);
}

int ssa_batch_from_blob(ssa_program_t* program, const uint8_t* blob, size_t size, const ArrowSchema* schema, ssa_batch_t** out, char** err) {
    return ssa_batches_from_blobs(program, &blob, &size, 1, schema, out, err);
}

int ssa_batch_from_device(const ssa_device_column* cols, size_t ncols, int64_t rows, ssa_batch_t** out, char** err) {
    if (err) *err = nullptr;
    auto batch = std::make_shared<Batch>();
    batch->Rows = rows;
    for (size_t i = 0; i < ncols; ++i) {
        const ssa_device_column& c = cols[i];
        Column col;
        col.Field.Name = c.name ? c.name : "";
        col.Field.Type = c.type;
        col.Length = rows;
        if (c.type < T_BOOL || c.type > T_DEC128) return Fail(Status::Invalid("bad column type"), err);
        if (c.validity && c.null_count != 0) {
            col.Validity = Buffer::Wrap(const_cast<uint8_t*>(c.validity), size_t((rows + 7) / 8));
            col.Field.Nullable = true;
        }
        col.NullCount = c.validity ? c.null_count : 0;
        if (c.index_type != 0) {  // dictionary-encoded: values = codes, chars / dict_offsets = the dictionary
            if (!IsStringLike(c.type)) return Fail(Status::Unsupported("dictionary of non-string values"), err);
            if (!IsInt(c.index_type)) return Fail(Status::Invalid("dictionary index type"), err);
            if (c.dict_size < 0 || (c.dict_size > 0 && !c.dict_offsets)) return Fail(Status::Invalid("dictionary without offsets"), err);
            auto dict = std::make_shared<DictInfo>();
            dict->IndexType = c.index_type;
            dict->Entries.Field.Type = c.type;
            dict->Entries.Field.Name = col.Field.Name;
            dict->Entries.Length = c.dict_size;
            dict->Entries.Values = Buffer::Wrap(const_cast<int32_t*>(c.dict_offsets), size_t(c.dict_size + 1) * 4);
            dict->Entries.Chars = Buffer::Wrap(const_cast<uint8_t*>(c.chars), size_t(c.chars_size) + 16);
            col.Values = Buffer::Wrap(const_cast<void*>(c.values), size_t(rows) * TypeWidth(c.index_type));
            col.Dict = dict;
            batch->Cols.push_back(std::move(col));
            continue;
        }
        const int w = IsStringLike(c.type) ? 4 : TypeWidth(c.type);
        size_t bytes = (c.type == T_BOOL) ? size_t((rows + 7) / 8) : size_t(rows + (IsStringLike(c.type) ? 1 : 0)) * w;
        col.Values = Buffer::Wrap(const_cast<void*>(c.values), bytes);
        if (IsStringLike(c.type)) col.Chars = Buffer::Wrap(const_cast<uint8_t*>(c.chars), size_t(c.chars_size) + 16);
        batch->Cols.push_back(std::move(col));
    }
    *out = new ssa_batch{batch};
    return SSA_OK;
}

void ssa_dictionary_stats(int64_t* steps_on_codes, int64_t* columns_decoded) {
    if (steps_on_codes) *steps_on_codes = ssa::gDictStepsOnCodes.load();
    if (columns_decoded) *columns_decoded = ssa::gDictColumnsDecoded.load();
}

int ssa_program_apply_device(ssa_program_t* program, const ssa_batch_t* in, ssa_batch_t** out, char** err) {
    if (err) *err = nullptr;
    std::lock_guard<std::mutex> g(program->Lock);
    {
        Status est = EnsureExecutor(program);
        if (!est.ok()) return Fail(est, err);
    }
    BatchPtr result;
    Status st = program->Exec->ApplyProgram(*program->Container.GetProgram(), in->Batch, result);
    if (!st.ok()) return Fail(st, err);
    *out = new ssa_batch{result};
    return SSA_OK;
}

int ssa_program_apply_many(ssa_program_t* program, const ssa_batch_t* const* in, size_t n, ssa_batch_t** out, char** err) {
    if (err) *err = nullptr;
    if (n > 0 && (!in || !out)) return Fail(Status::Invalid("null batch array"), err);
    std::lock_guard<std::mutex> g(program->Lock);
    {
        Status est = EnsureExecutor(program);
        if (!est.ok()) return Fail(est, err);
    }
    std::vector<BatchPtr> batches, results;
    for (size_t i = 0; i < n; ++i) {
        if (!in[i]) return Fail(Status::Invalid("null batch"), err);
        batches.push_back(in[i]->Batch);
    }
    Status st = program->Exec->ApplyProgramMany(*program->Container.GetProgram(), batches, results);
    if (!st.ok()) return Fail(st, err);
    for (size_t i = 0; i < n; ++i) out[i] = new ssa_batch{results[i]};
    return SSA_OK;
}

int ssa_batch_pack_layout(ssa_program_t* program, const ssa_batch_t* batch, uint8_t* header, size_t header_capacity, size_t* header_bytes,
                          size_t* data_bytes, char** err) {
    if (err) *err = nullptr;
    std::lock_guard<std::mutex> g(program->Lock);
    {
        Status est = EnsureExecutor(program);
        if (!est.ok()) return Fail(est, err);
    }
    std::vector<uint8_t> h;
    uint64_t data = 0;
    Status st = program->Exec->PackedLayout(*batch->Batch, h, data);
    if (!st.ok()) return Fail(st, err);
    if (header_bytes) *header_bytes = h.size();
    if (data_bytes) *data_bytes = size_t(data);
    if (header) {
        if (h.size() > header_capacity) return Fail(Status::Invalid("the header does not fit: " + std::to_string(h.size()) + " bytes"), err);
        std::memcpy(header, h.data(), h.size());
    }
    return SSA_OK;
}

int ssa_batch_pack_data(ssa_program_t* program, const ssa_batch_t* batch, void* device_dst, size_t capacity, char** err) {
    if (err) *err = nullptr;
    std::lock_guard<std::mutex> g(program->Lock);
    {
        Status est = EnsureExecutor(program);
        if (!est.ok()) return Fail(est, err);
    }
    Status st = program->Exec->PackData(*batch->Batch, device_dst, capacity);
    if (!st.ok()) return Fail(st, err);
    return SSA_OK;
}

int ssa_batch_unpack(const uint8_t* header, size_t header_bytes, void* device_data, size_t data_bytes, ssa_batch_t** out, char** err) {
    if (err) *err = nullptr;
    BatchPtr b;
    Status st = UnpackBatch(header, header_bytes, device_data, data_bytes, b);
    if (!st.ok()) return Fail(st, err);
    *out = new ssa_batch{b};
    return SSA_OK;
}

int ssa_batch_concat(ssa_program_t* program, const ssa_batch_t* const* parts, size_t n, ssa_batch_t** out, char** err) {
    if (err) *err = nullptr;
    if (n == 0 || !parts) return Fail(Status::Invalid("no batches"), err);
    std::lock_guard<std::mutex> g(program->Lock);
    {
        Status est = EnsureExecutor(program);
        if (!est.ok()) return Fail(est, err);
    }
    std::vector<BatchPtr> in;
    for (size_t i = 0; i < n; ++i) {
        if (!parts[i]) return Fail(Status::Invalid("null batch"), err);
        in.push_back(parts[i]->Batch);
    }
    BatchPtr res;
    Status st = program->Exec->ConcatBatches(in, res);
    if (!st.ok()) return Fail(st, err);
    *out = new ssa_batch{res};
    return SSA_OK;
}

int ssa_batch_slice(const ssa_batch_t* batch, int64_t row0, int64_t rows, ssa_batch_t** out, char** err) {
    if (err) *err = nullptr;
    const Batch& b = *batch->Batch;
    if (row0 < 0 || rows < 0 || row0 + rows > b.Rows) return Fail(Status::Invalid("rows outside the batch"), err);
    auto v = std::make_shared<Batch>();
    v->Rows = rows;
    for (const Column& c : b.Cols) v->Cols.push_back(ColumnView(c, row0, rows));
    *out = new ssa_batch{v};
    return SSA_OK;
}

int ssa_batch_rename(const ssa_batch_t* batch, const char* const* names, size_t n, ssa_batch_t** out, char** err) {
    if (err) *err = nullptr;
    if (n != batch->Batch->Cols.size()) return Fail(Status::Invalid("one name per column is needed"), err);
    auto v = std::make_shared<Batch>(*batch->Batch);
    for (size_t i = 0; i < n; ++i) v->Cols[i].Field.Name = names[i] ? names[i] : "";
    *out = new ssa_batch{v};
    return SSA_OK;
}

char* ssa_batch_column_name(const ssa_batch_t* batch, size_t i) {
    if (i >= batch->Batch->Cols.size()) return nullptr;
    return DupString(batch->Batch->Cols[i].Field.Name);
}

int ssa_merge_sorted(ssa_program_t* program, const ssa_batch_t* const* sources, const ssa_filter_t* const* filters, size_t n_sources,
                     const char* const* key_columns, size_t n_keys, const char* const* version_columns, size_t n_versions, int reverse,
                     ssa_batch_t** out, char** err) {
    if (err) *err = nullptr;
    if (!sources || n_sources == 0) return Fail(Status::Invalid("no sources"), err);
    std::lock_guard<std::mutex> g(program->Lock);
    {
        Status est = EnsureExecutor(program);
        if (!est.ok()) return Fail(est, err);
    }
    std::vector<BatchPtr> in;
    std::vector<FilterPtr> fs;
    for (size_t i = 0; i < n_sources; ++i) {
        if (!sources[i]) return Fail(Status::Invalid("null source"), err);
        in.push_back(sources[i]->Batch);
        if (filters) fs.push_back(filters[i] ? filters[i]->Filter : nullptr);
    }
    std::vector<std::string> keys, versions;
    for (size_t i = 0; i < n_keys; ++i) keys.emplace_back(key_columns[i] ? key_columns[i] : "");
    for (size_t i = 0; i < n_versions; ++i) versions.emplace_back(version_columns[i] ? version_columns[i] : "");
    BatchPtr res;
    Status st = program->Exec->MergeSorted(in, fs, keys, versions, reverse != 0, res);
    if (!st.ok()) return Fail(st, err);
    *out = new ssa_batch{res};
    return SSA_OK;
}

void ssa_shutdown(void) { ssa::KernelJit::Instance().Shutdown(); }

void ssa_apply_many_stats(int64_t* steps_in_one_launch, int64_t* steps_slice_by_slice) {
    if (steps_in_one_launch) *steps_in_one_launch = ssa::gManyStepsOneLaunch.load();
    if (steps_slice_by_slice) *steps_slice_by_slice = ssa::gManyStepsPerSlice.load();
}

void ssa_keyed_compact_stats(int64_t* scans_with_compacted_tiles) {
    if (scans_with_compacted_tiles) *scans_with_compacted_tiles = ssa::gKeyedStepsCompacted.load();
}

void ssa_keyed_stats(int64_t* scans_applied_by_region) {
    if (scans_applied_by_region) *scans_applied_by_region = ssa::gKeyedStepsDeferred.load();
}

void ssa_keyed_sort_stats(int64_t* sorts_done_again_with_counts) {
    if (sorts_done_again_with_counts) *sorts_done_again_with_counts = ssa::gRadixCursorOverflows.load();
}

int ssa_batch_export(ssa_program_t* program, const ssa_batch_t* batch, ArrowArray* out, ArrowSchema* out_schema, char** err) {
    if (err) *err = nullptr;
    std::lock_guard<std::mutex> g(program->Lock);
    {
        Status est = EnsureExecutor(program);
        if (!est.ok()) return Fail(est, err);
    }
    Status st = program->Exec->ExportBatch(*batch->Batch, out, out_schema);
    if (!st.ok()) return Fail(st, err);
    return SSA_OK;
}

int ssa_sharding_hash(ssa_program_t* program, const ssa_batch_t* batch, const char* const* key_columns, size_t n_keys, uint64_t seed,
                      uint64_t* out_hashes, char** err) {
    if (err) *err = nullptr;
    std::lock_guard<std::mutex> g(program->Lock);
    {
        Status est = EnsureExecutor(program);
        if (!est.ok()) return Fail(est, err);
    }
    std::vector<std::string> names;
    for (size_t i = 0; i < n_keys; ++i) names.emplace_back(key_columns[i] ? key_columns[i] : "");
    BufferPtr hashes;
    BatchPtr plain;  // dictionary columns as the strings they stand for
    Status st = program->Exec->DecodedBatch(batch->Batch, plain);
    if (st.ok()) st = program->Exec->ShardingHash(*plain, names, seed, hashes);
    if (!st.ok()) return Fail(st, err);
    if (batch->Batch->Rows > 0 && out_hashes) {
        if (hipMemcpyAsync(out_hashes, hashes->Ptr, size_t(batch->Batch->Rows) * 8, hipMemcpyDeviceToHost, program->Exec->Stream()) != hipSuccess ||
            hipStreamSynchronize(program->Exec->Stream()) != hipSuccess) {
            return Fail(Status::Hip("copying the hashes to the host failed"), err);
        }
    }
    return SSA_OK;
}

namespace {
int ShardingSplitEntry(ssa_program_t* program, const ssa_batch_t* batch, const char* const* key_columns, size_t n_keys, uint32_t shards_count,
                       bool balance, ssa_batch_t** out, uint32_t* out_shard_rows, char** err);
}

int ssa_sharding_split(ssa_program_t* program, const ssa_batch_t* batch, const char* const* key_columns, size_t n_keys,
                       uint32_t shards_count, ssa_batch_t** out, uint32_t* out_shard_rows, char** err) {
    return ShardingSplitEntry(program, batch, key_columns, n_keys, shards_count, /*balance=*/true, out, out_shard_rows, err);
}

int ssa_batch_partition_by_key(ssa_program_t* program, const ssa_batch_t* batch, const char* const* key_columns, size_t n_keys,
                               uint32_t parts_count, ssa_batch_t** out, uint32_t* out_part_rows, char** err) {
    return ShardingSplitEntry(program, batch, key_columns, n_keys, parts_count, /*balance=*/false, out, out_part_rows, err);
}

namespace {
int ShardingSplitEntry(ssa_program_t* program, const ssa_batch_t* batch, const char* const* key_columns, size_t n_keys, uint32_t shards_count,
                       bool balance, ssa_batch_t** out, uint32_t* out_shard_rows, char** err) {
    if (err) *err = nullptr;
    std::lock_guard<std::mutex> g(program->Lock);
    {
        Status est = EnsureExecutor(program);
        if (!est.ok()) return Fail(est, err);
    }
    std::vector<std::string> names;
    for (size_t i = 0; i < n_keys; ++i) names.emplace_back(key_columns[i] ? key_columns[i] : "");
    BatchPtr result;
    std::vector<uint32_t> rows;
    BatchPtr plain;
    Status st = program->Exec->DecodedBatch(batch->Batch, plain);
    if (st.ok()) st = program->Exec->ShardingSplit(*plain, names, shards_count, result, rows, balance);
    if (!st.ok()) return Fail(st, err);
    if (out_shard_rows) std::memcpy(out_shard_rows, rows.data(), rows.size() * sizeof(uint32_t));
    *out = new ssa_batch{result};
    return SSA_OK;
}
}  // namespace

// ---- TColumnFilter ----------------------------------------------------------------------------------------

int ssa_filter_simple(int allow, ssa_filter_t** out) {
    auto f = std::make_shared<DeviceFilter>();
    f->Empty = true;
    f->DefaultValue = allow != 0;
    *out = new ssa_filter{f};
    return SSA_OK;
}

#define SSA_FILTER_ENTER()                                \
    if (err) *err = nullptr;                              \
    std::lock_guard<std::mutex> g(program->Lock);         \
    {                                                     \
        Status est = EnsureExecutor(program);             \
        if (!est.ok()) return Fail(est, err);             \
    }

int ssa_filter_from_runs(ssa_program_t* program, int first_value, const uint32_t* runs, size_t n_runs, ssa_filter_t** out, char** err) {
    SSA_FILTER_ENTER();
    FilterPtr f;
    Status st = program->Exec->FilterFromRuns(first_value != 0, std::vector<uint32_t>(runs, runs + n_runs), f);
    if (!st.ok()) return Fail(st, err);
    *out = new ssa_filter{f};
    return SSA_OK;
}

int ssa_filter_from_program(ssa_program_t* program, size_t step_index, const ssa_batch_t* batch, ssa_filter_t** out, char** err) {
    SSA_FILTER_ENTER();
    auto& steps = program->Container.GetProgram()->Steps;
    if (step_index >= steps.size()) return Fail(Status::Invalid("no such step"), err);
    *out = nullptr;
    if (steps[step_index]->Filters.empty()) return SSA_OK;
    FilterPtr f;
    Status st = program->Exec->FilterFromStep(*steps[step_index], batch->Batch, f);
    if (!st.ok()) return Fail(st, err);
    *out = new ssa_filter{f};
    return SSA_OK;
}

int ssa_filter_from_program_noalloc(ssa_program_t* program, size_t step_index, const ssa_batch_t* batch, uint64_t* out_bits,
                                    int64_t* out_selected, char** err) {
    SSA_FILTER_ENTER();
    auto& steps = program->Container.GetProgram()->Steps;
    if (step_index >= steps.size()) return Fail(Status::Invalid("no such step"), err);
    if (steps[step_index]->Filters.empty()) return Fail(Status::Invalid("the step has no Filter command"), err);
    if (!batch || !out_selected) return Fail(Status::Invalid("null argument"), err);
    int64_t selected = 0;
    Status st = program->Exec->FilterFromStepNoAlloc(*steps[step_index], *batch->Batch, out_bits, selected);
    if (!st.ok()) return Fail(st, err);
    *out_selected = selected;
    return SSA_OK;
}

int ssa_filter_predicate(ssa_program_t* program, const ssa_batch_t* batch, const ArrowArray* border, const ArrowSchema* border_schema,
                         int compare_type, ssa_filter_t** out, char** err) {
    SSA_FILTER_ENTER();
    if (!border || !border_schema) return Fail(Status::Invalid("null border"), err);
    if (border->length != 1) return Fail(Status::Invalid("the border must have exactly one row"), err);
    std::vector<std::pair<std::string, TScalar>> row;
    Status st = ReadRowScalars(border, border_schema, row);
    if (!st.ok()) return Fail(st, err);
    if (int64_t(row.size()) != border->n_children) return Fail(Status::Unsupported("border column type"), err);
    std::vector<BorderValue> keys;
    for (auto& kv : row) keys.push_back(BorderValue{kv.first, kv.second});
    FilterPtr f;
    BatchPtr plain;
    st = program->Exec->DecodedBatch(batch->Batch, plain);
    if (st.ok()) st = program->Exec->FilterPredicate(*plain, keys, compare_type, f);
    if (!st.ok()) return Fail(st, err);
    *out = new ssa_filter{f};
    return SSA_OK;
}

static int FilterMerge(ssa_program_t* program, const ssa_filter_t* a, const ssa_filter_t* b, bool isOr, ssa_filter_t** out, char** err) {
    SSA_FILTER_ENTER();
    FilterPtr f;
    Status st = program->Exec->FilterAndOr(*a->Filter, *b->Filter, isOr, f);
    if (!st.ok()) return Fail(st, err);
    *out = new ssa_filter{f};
    return SSA_OK;
}
int ssa_filter_and(ssa_program_t* program, const ssa_filter_t* a, const ssa_filter_t* b, ssa_filter_t** out, char** err) {
    return FilterMerge(program, a, b, false, out, err);
}
int ssa_filter_or(ssa_program_t* program, const ssa_filter_t* a, const ssa_filter_t* b, ssa_filter_t** out, char** err) {
    return FilterMerge(program, a, b, true, out, err);
}

int ssa_filter_combine_sequential_and(ssa_program_t* program, const ssa_filter_t* self, const ssa_filter_t* ext, ssa_filter_t** out,
                                      char** err) {
    SSA_FILTER_ENTER();
    FilterPtr f;
    Status st = program->Exec->FilterCombineSequentialAnd(*self->Filter, *ext->Filter, f);
    if (!st.ok()) return Fail(st, err);
    *out = new ssa_filter{f};
    return SSA_OK;
}

void ssa_filter_info(const ssa_filter_t* filter, int* is_empty, int* default_value, int64_t* count) {
    if (is_empty) *is_empty = filter->Filter->Empty ? 1 : 0;
    if (default_value) *default_value = filter->Filter->DefaultValue ? 1 : 0;
    if (count) *count = filter->Filter->Empty ? 0 : filter->Filter->Count;
}

int ssa_filter_runs(ssa_program_t* program, const ssa_filter_t* filter, uint32_t* runs, size_t* n_runs, int* last_value, char** err) {
    SSA_FILTER_ENTER();
    std::vector<uint32_t> r;
    bool last = true;
    Status st = program->Exec->FilterToRuns(*filter->Filter, r, last);
    if (!st.ok()) return Fail(st, err);
    if (runs) {
        if (!n_runs || *n_runs < r.size()) return Fail(Status::Invalid("the runs array is too small"), err);
        std::memcpy(runs, r.data(), r.size() * sizeof(uint32_t));
    }
    if (n_runs) *n_runs = r.size();
    if (last_value) *last_value = last ? 1 : 0;
    return SSA_OK;
}

int ssa_filter_bits(ssa_program_t* program, const ssa_filter_t* filter, uint8_t* out_bits, char** err) {
    SSA_FILTER_ENTER();
    std::vector<uint8_t> bits;
    Status st = program->Exec->FilterBits(*filter->Filter, bits);
    if (!st.ok()) return Fail(st, err);
    if (!bits.empty()) std::memcpy(out_bits, bits.data(), bits.size());
    return SSA_OK;
}

int ssa_filter_filtered_count(ssa_program_t* program, const ssa_filter_t* filter, int64_t* out, char** err) {
    SSA_FILTER_ENTER();
    int64_t n = 0;
    Status st = program->Exec->FilterCountSelected(*filter->Filter, n);
    if (!st.ok()) return Fail(st, err);
    if (out) *out = n;
    return SSA_OK;
}

int ssa_filter_apply(ssa_program_t* program, const ssa_filter_t* filter, const ssa_batch_t* batch, ssa_batch_t** out, char** err) {
    SSA_FILTER_ENTER();
    BatchPtr result;
    Status st = program->Exec->FilterApply(*filter->Filter, batch->Batch, result);
    if (!st.ok()) return Fail(st, err);
    *out = new ssa_batch{result};
    return SSA_OK;
}

void ssa_filter_destroy(ssa_filter_t* filter) { delete filter; }

int64_t ssa_batch_rows(const ssa_batch_t* batch) { return batch->Batch->Rows; }
size_t ssa_batch_columns(const ssa_batch_t* batch) { return batch->Batch->Cols.size(); }
void ssa_batch_destroy(ssa_batch_t* batch) { delete batch; }

void ssa_profile_enable(ssa_program_t* program, int enabled) {
    std::lock_guard<std::mutex> g(program->Lock);
    if (!EnsureExecutor(program).ok()) return;
    program->Exec->ProfileCollect();
    program->Exec->Profile.Enabled = enabled != 0;
}

void ssa_specialization_stats(int64_t* kernels_built, double* build_ms, int64_t* launches) {
    int64_t built = 0, n = 0;
    double ms = 0;
    ssa::KernelJit::Instance().Stats(built, ms, n);
    if (kernels_built) *kernels_built = built;
    if (build_ms) *build_ms = ms;
    if (launches) *launches = n;
}

void ssa_profile_read(ssa_program_t* program, double* scan_ms, int64_t* launches, int64_t* algorithmic_bytes) {
    std::lock_guard<std::mutex> g(program->Lock);
    if (scan_ms) *scan_ms = 0;
    if (launches) *launches = 0;
    if (algorithmic_bytes) *algorithmic_bytes = 0;
    if (!program->Exec) return;
    program->Exec->ProfileCollect();
    KernelProfile& p = program->Exec->Profile;
    if (scan_ms) *scan_ms = p.ScanMs;
    if (launches) *launches = p.ScanLaunches;
    if (algorithmic_bytes) *algorithmic_bytes = p.ScanBytes;
    p.ScanMs = 0;
    p.ScanLaunches = 0;
    p.ScanBytes = 0;
}

}  // extern "C"
