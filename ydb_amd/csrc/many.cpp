// A program over the slices of a table, one result per slice.
//
// The reference applies a program to one record batch at a time (NSsa::TProgram::ApplyTo per slice,
// ydb/core/formats/arrow/program.h:423-433); at its batch sizes a GPU scan of one slice is over in microseconds and
// the launches, allocations and host round trips around it dominate.  Here the slices of one schema whose step is an
// aggregation are scanned by one launch: the grid is divided among the slices in proportion to their rows, every
// workgroup reads its own slice's columns and aggregates into its slice's part of the hash table, and the groups leave
// the table slice after slice -- the result of slice i is a view of rows [start_i, start_i+1) of one set of output
// columns.  Steps the one-launch path does not cover run slice by slice.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <map>

#include "exec.h"

namespace ssa {

std::atomic<int64_t> gManyStepsOneLaunch{0};    // steps of ApplyProgramMany that scanned all their slices with one launch
std::atomic<int64_t> gManyStepsPerSlice{0};     // steps that ran slice by slice

#define SSA_HIP(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            (void)hipGetLastError();                                                                   \
            return Status::Hip(std::string(#expr) + ": " + hipGetErrorString(e_));                     \
        }                                                                                              \
    } while (0)

#define SSA_RETURN_NOT_OK(expr)     \
    do {                            \
        Status s_ = (expr);         \
        if (!s_.ok()) return s_;    \
    } while (0)

Status Executor::ApplyProgramMany(const TProgram& program, const std::vector<BatchPtr>& in, std::vector<BatchPtr>& out) {
    HostPhases::Mark("begin");
    (void)hipSetDevice(Device);
    std::vector<BatchPtr> cur = in;
    for (auto& step : program.Steps) {
        std::vector<BatchPtr> next(cur.size());
        // slices that agree in which columns carry nulls share a plan: one launch per such set
        std::map<std::string, std::vector<size_t>> sets;
        for (size_t i = 0; i < cur.size(); ++i) {
            if (!cur[i]) return Status::Invalid("null batch");
            std::string sig;
            for (const Column& c : cur[i]->Cols) sig += c.Validity ? '1' : '0';
            sets[sig].push_back(i);
        }
        for (auto& kv : sets) {
            const std::vector<size_t>& members = kv.second;
            Status st = Status::Unsupported("one slice");
            if (members.size() > 1) {
                std::vector<BatchPtr> part, res;
                for (size_t i : members) part.push_back(cur[i]);
                st = ApplyStepMany(*step, part, res);
                if (st.ok()) {
                    gManyStepsOneLaunch.fetch_add(1);
                    for (size_t k = 0; k < members.size(); ++k) next[members[k]] = res[k];
                }
            }
            if (st.Code == ST_UNSUPPORTED) {
                gManyStepsPerSlice.fetch_add(1);
                for (size_t i : members) SSA_RETURN_NOT_OK(ApplyStep(*step, cur[i], next[i]));
            } else if (!st.ok()) {
                return st;
            }
        }
        cur = next;
    }
    out = cur;
    HostPhases::Mark("end");
    HostPhases::Report("apply of slices:");
    return Status::OK();
}

Status Executor::ApplyStepMany(const TProgramStep& step, const std::vector<BatchPtr>& in, std::vector<BatchPtr>& out) {
    const size_t n = in.size();
    const std::vector<FieldDesc> schema = in[0]->Schema();
    for (const BatchPtr& b : in) {
        if (!b) return Status::Invalid("null batch");
        if (b->Cols.size() != schema.size()) return Status::Unsupported("slices in one launch: differing schemas");
        for (size_t c = 0; c < schema.size(); ++c) {
            const Column& col = b->Cols[c];
            if (col.Dict) return Status::Unsupported("slices in one launch: dictionary-encoded columns");
            if (col.Field.Name != schema[c].Name || col.Field.Type != schema[c].Type || (col.Validity != nullptr) != schema[c].Nullable ||
                col.Field.IsScalar || col.Field.Precision != schema[c].Precision || col.Field.Scale != schema[c].Scale) {
                return Status::Unsupported("slices in one launch: differing schemas");
            }
        }
    }
    std::shared_ptr<const StepPlan> planKept;
    SSA_RETURN_NOT_OK(PlanFor(step, *in[0], /*stable=*/true, planKept));
    StableStep = true;
    const StepPlan& plan = *planKept;
    const bool noKey = plan.Kind == StepPlan::AGG_NOKEY;
    if (noKey) {
        // as one group per slice in the keyed sink: sums, counts and integer extremes are the same there
        if (plan.HasStringMinMax) return Status::Unsupported("slices in one launch: string MIN / MAX");
        for (const AggPlan& ap : plan.Aggs) {
            const int k = ap.Spec.kind;
            if (k != AGG_COUNT && k != AGG_SUM_I && k != AGG_SUM_F && k != AGG_MIN_I && k != AGG_MAX_I && k != AGG_MIN_U && k != AGG_MAX_U) {
                return Status::Unsupported("slices in one launch: this key-less aggregate");
            }
        }
    } else if (plan.Kind != StepPlan::AGG_KEYED) {
        return Status::Unsupported("slices in one launch: not an aggregation");
    }

    HostPhases::Mark("plan");
    ManySlices many;
    std::vector<Batch> shadows;  // the slices with their one-length string keys as fixed-width columns
    std::shared_ptr<const StepPlan> fixedKept;
    const StepPlan* usePlan = &plan;
    struct Rewrite {
        size_t Key;
        int Column;
        uint32_t Len;
        std::string Shadow;
    };
    std::vector<Rewrite> rewrites;
    bool stringKeys = false;
    for (const KeyPlan& kp : plan.Keys) stringKeys = stringKeys || IsStringLike(kp.Spec.type);
    if (stringKeys) {
        // Executor::RunKeyedUniform for every slice at once: string keys whose strings all have one length of 1, 2, 4 or
        // 8 bytes become fixed-width keys (the scan checks every offset); anything else runs slice by slice.
        for (size_t k = 0; k < plan.Keys.size(); ++k) {
            const KeyPlan& kp = plan.Keys[k];
            if (kp.InputColumn < 0) return Status::Unsupported("slices in one launch: computed key next to string keys");
            if (IsStringLike(kp.Spec.type)) rewrites.push_back(Rewrite{k, kp.InputColumn, 0, std::string("\x01fixed:") + std::to_string(k)});
        }
        if (rewrites.size() > size_t(kMaxOffsetChecks) || schema.size() + rewrites.size() > 64) return Status::Unsupported("slices in one launch: string keys");
        if (std::getenv("SSA_NO_UNIFORM_KEYS")) return Status::Unsupported("slices in one launch: string keys");
        // first, second and last offset of every such column of every slice, fetched together
        const size_t nprobe = n * rewrites.size() * 3;
        // (addresses and answers in pinned memory the launch reads and writes where it lies: no copies, no allocation)
        const bool pinnedProbe = nprobe * 12 + 64 <= kHostStagingBytes && EnsureHostStaging();
        std::vector<const int32_t*> pageable(pinnedProbe ? 0 : nprobe);
        const int32_t** addrs = pinnedProbe ? reinterpret_cast<const int32_t**>(HostStaging) : pageable.data();
        for (size_t i = 0; i < n; ++i) {
            if (in[i]->Rows < 1) return Status::Unsupported("slices in one launch: an empty slice with string keys");
            for (size_t r = 0; r < rewrites.size(); ++r) {
                const Column& col = in[i]->Cols[size_t(rewrites[r].Column)];
                if (!col.Values || !col.Chars) return Status::Unsupported("slices in one launch: string keys");
                const int32_t* offs = static_cast<const int32_t*>(col.ValuesPtr());
                if (reinterpret_cast<uintptr_t>(offs) & 15) return Status::Unsupported("slices in one launch: unaligned offsets");
                const size_t at = (i * rewrites.size() + r) * 3;
                addrs[at] = offs;
                addrs[at + 1] = offs + 1;
                addrs[at + 2] = offs + in[i]->Rows;
            }
        }
        std::vector<int32_t> answers(pinnedProbe ? 0 : nprobe);
        const int32_t* probe = answers.data();
        if (pinnedProbe) {
            int32_t* out = reinterpret_cast<int32_t*>(HostStaging + ((nprobe * 8 + 63) & ~size_t(63)));
            SSA_HIP(LaunchGatherI32At(addrs, int64_t(nprobe), out, StreamHandle));
            HostPhases::Mark("probe-launch");
            SSA_HIP(hipStreamSynchronize(StreamHandle));
            probe = out;
        } else {
            BufferPtr dAddrs = Buffer::Alloc(nprobe * 8 + 64), dProbe = Buffer::Alloc(nprobe * 4 + 64);
            if (!dAddrs || !dProbe) return Status::Hip("out of device memory");
            SSA_HIP(hipMemcpyAsync(dAddrs->Ptr, addrs, nprobe * 8, hipMemcpyHostToDevice, StreamHandle));
            SSA_HIP(LaunchGatherI32At(static_cast<const int32_t* const*>(dAddrs->Ptr), int64_t(nprobe), static_cast<int32_t*>(dProbe->Ptr), StreamHandle));
            SSA_HIP(hipMemcpyAsync(answers.data(), dProbe->Ptr, nprobe * 4, hipMemcpyDeviceToHost, StreamHandle));
            HostPhases::Mark("probe-launch");
            SSA_HIP(hipStreamSynchronize(StreamHandle));
        }
        HostPhases::Mark("probe-sync");
        shadows.resize(n);
        many.Checks.resize(n);
        for (size_t i = 0; i < n; ++i) {
            shadows[i] = *in[i];
            for (size_t r = 0; r < rewrites.size(); ++r) {
                const size_t at = (i * rewrites.size() + r) * 3;
                const int64_t first = probe[at], len = int64_t(probe[at + 1]) - first, last = probe[at + 2];
                const Column& src = in[i]->Cols[size_t(rewrites[r].Column)];
                if (first < 0 || (len != 1 && len != 2 && len != 4 && len != 8) || last - first != len * in[i]->Rows || size_t(last) > src.Chars->Size) {
                    return Status::Unsupported("slices in one launch: string keys of differing lengths");
                }
                if (i == 0) rewrites[r].Len = uint32_t(len);
                if (rewrites[r].Len != uint32_t(len)) return Status::Unsupported("slices in one launch: string keys of differing lengths");
                Column col;
                col.Field.Name = rewrites[r].Shadow;
                col.Field.Type = len == 1 ? T_U8 : len == 2 ? T_U16 : len == 4 ? T_U32 : T_U64;
                col.Length = in[i]->Rows;
                col.Values = Buffer::Wrap(static_cast<char*>(src.Chars->Ptr) + first, size_t(in[i]->Rows) * size_t(len));
                shadows[i].Cols.push_back(col);
                many.Checks[i].push_back(OffsetCheck{static_cast<const int32_t*>(src.ValuesPtr()), uint32_t(first), uint32_t(len)});
            }
        }
        int keyBits = 0;
        for (const KeyPlan& kp : plan.Keys) {
            if (!IsStringLike(kp.Spec.type)) keyBits += 8 * TypeWidth(kp.Spec.type);
        }
        for (const Rewrite& r : rewrites) keyBits += int(8 * r.Len);
        if (keyBits > 119) return Status::Unsupported("slices in one launch: wide keys");
        std::vector<std::pair<size_t, std::string>> renames;
        std::vector<uint32_t> lens;
        for (const Rewrite& r : rewrites) {
            renames.emplace_back(r.Key, r.Shadow);
            lens.push_back(r.Len);
        }
        if (!FixedKeysPlan(step, shadows[0], renames, lens, fixedKept).ok() || fixedKept->Kind != StepPlan::AGG_KEYED) {
            return Status::Unsupported("slices in one launch: string keys");
        }
        usePlan = fixedKept.get();
        for (size_t i = 0; i < n; ++i) many.In.push_back(&shadows[i]);
    } else {
        for (size_t i = 0; i < n; ++i) many.In.push_back(in[i].get());
    }

    HostPhases::Mark("shadows+plan");
    BatchPtr all;
    bool notUniform = false;
    static const std::vector<OffsetCheck> kNone;
    SSA_RETURN_NOT_OK(RunKeyedScan(step, *usePlan, *many.In[0], all, many.Checks.empty() ? kNone : many.Checks[0], &notUniform, &many));
    if (notUniform) return Status::Unsupported("slices in one launch: string keys of differing lengths");
    if (many.GroupStart.size() != n + 1) return Status::Invalid("internal: group ranges of the slices");

    // one-length string keys back as strings: the emitted values are the bytes, the offsets multiples of the length
    for (Column& col : all->Cols) {
        for (const Rewrite& r : rewrites) {
            if (col.Field.Name != r.Shadow) continue;
            const Column& src = in[0]->Cols[size_t(r.Column)];
            Column str;
            str.Field = src.Field;
            str.Field.Nullable = false;
            str.Length = all->Rows;
            SSA_RETURN_NOT_OK(OffsetRamp(int(r.Len), all->Rows, str.Values));
            str.Chars = col.Values;
            if (!str.Chars) {
                str.Chars = Buffer::Alloc(16);
                if (!str.Chars) return Status::Hip("out of device memory");
            }
            col = str;
        }
    }

    HostPhases::Mark("scan-returned");
    out.assign(n, nullptr);
    BatchPtr emptyRow;  // key-less aggregation over no row: nulls, zero counts
    for (size_t i = 0; i < n; ++i) {
        const int64_t start = many.GroupStart[i], count = many.GroupCount.empty() ? many.GroupStart[i + 1] - start : many.GroupCount[i];
        if (noKey && count != 1) {
            if (count != 0) return Status::Invalid("internal: key-less aggregation produced more than one group");
            if (!emptyRow) {
                emptyRow = std::make_shared<Batch>();
                emptyRow->Rows = 1;
                for (const OutColumn& oc : plan.Outs) {
                    Column col;
                    col.Field.Name = oc.Name;
                    col.Field.Type = oc.Type;
                    col.Field.Precision = oc.Precision;
                    col.Field.Scale = oc.Scale;
                    col.Length = 1;
                    col.Values = Buffer::Alloc(64);
                    if (!col.Values) return Status::Hip("out of device memory");
                    SSA_HIP(hipMemsetAsync(col.Values->Ptr, 0, 64, StreamHandle));
                    if (plan.Aggs[size_t(oc.Index)].Spec.kind != AGG_COUNT) {
                        col.Validity = Buffer::Alloc(32);
                        if (!col.Validity) return Status::Hip("out of device memory");
                        SSA_HIP(hipMemsetAsync(col.Validity->Ptr, 0, 32, StreamHandle));
                        col.NullCount = 1;
                        col.Field.Nullable = true;
                    }
                    emptyRow->Cols.push_back(std::move(col));
                }
            }
            out[i] = emptyRow;
            continue;
        }
        auto b = std::make_shared<Batch>();
        b->Rows = count;
        for (const Column& col : all->Cols) b->Cols.push_back(ColumnView(col, start, count));
        out[i] = b;
    }
    HostPhases::Mark("views");
    SSA_HIP(hipStreamSynchronize(StreamHandle));
    return Status::OK();
}

}  // namespace ssa
