// CPU oracle of NArrow::TColumnFilter's run-length algebra.  TEST INFRASTRUCTURE ONLY: nothing in the product links,
// loads or calls this file; tests/ and the smoke check use it as the checker of the GPU filter kernels.
//
// A restatement, in this build's own words, of ydb/core/formats/arrow/arrow_filter.{h,cpp} of the reference: the filter is
// a vector of run lengths whose values alternate, plus LastValue / Count / DefaultFilterValue.  The walks over two
// run lists below follow the reference step for step (cited per function) so that the produced run vectors - not just
// the bits they stand for - can be compared with what the GPU path reports.
//
// Pinned by the reference's own known answers: ut_column_filter.cpp:9-27 (Or, CombineSequentialAnd) and
// ut_arrow.cpp:604-634 (KeyComparison: 1000 rows against the border {2,3,4}); see tests/test_oracle_filter.py.
#include <algorithm>
#include <cstdint>
#include <vector>

namespace {

struct RleFilter {
    bool Default = true;   // DefaultFilterValue (arrow_filter.h:22)
    bool Last = true;      // LastValue (:23)
    uint32_t Count = 0;    // :24
    std::vector<uint32_t> Runs;  // Filter (:25)

    // GetStartValue (arrow_filter.h:34-47), forward direction
    bool Start() const {
        if (Runs.empty()) return Default;
        return (Runs.size() % 2 == 0) ? !Last : Last;
    }
    // Add (arrow_filter.cpp:239-250)
    void Add(bool value, uint32_t n) {
        if (n == 0) return;
        if (Last != value || Count == 0) {
            Runs.push_back(n);
            Last = value;
        } else {
            Runs.back() += n;
        }
        Count += n;
    }
    // IsTotalAllowFilter / IsTotalDenyFilter (arrow_filter.cpp:212-230)
    bool TotalAllow() const { return (Default && Runs.empty()) || (Runs.size() == 1 && Last); }
    bool TotalDeny() const { return (!Default && Runs.empty()) || (Runs.size() == 1 && !Last); }
};

RleFilter* Simple(bool value) {
    auto* f = new RleFilter();
    f->Default = value;
    return f;
}

// CrossSize (arrow_filter.cpp:252-257): length of the overlap of [s1, f1) and [s2, f2)
bool Overlap(uint32_t s1, uint32_t f1, uint32_t s2, uint32_t f2, uint32_t& out) {
    const uint32_t f = std::min(f1, f2), s = std::max(s1, s2);
    if (f < s) return false;  // the reference aborts
    out = f - s;
    return true;
}

// The accumulation both walks share: extend the last run or open a new one (arrow_filter.cpp:468-476, 537-545).
struct RunSink {
    std::vector<uint32_t> Runs;
    bool Current = false;
    uint32_t Count = 0;
    void Put(bool value, uint32_t n) {
        if (n == 0) return;
        if (Count == 0 || Current != value) {
            Runs.push_back(n);
            Current = value;
        } else {
            Runs.back() += n;
        }
        Count += n;
    }
    RleFilter* Finish() {
        auto* f = new RleFilter();  // BuildAllowFilter, then Filter / LastValue / Count swapped in (:495-499)
        f->Runs.swap(Runs);
        f->Last = Current;
        f->Count = Count;
        return f;
    }
};

// TMergerImpl::Merge<TMergePolicy> (arrow_filter.cpp:444-501)
RleFilter* Merge(const RleFilter& a, const RleFilter& b, bool isOr) {
    auto calc = [&](bool x, bool y) { return isOr ? (x || y) : (x && y); };
    // MergeWithSimple (:408-429)
    auto withSimple = [&](const RleFilter& sized, bool simple) -> RleFilter* {
        if (isOr) return simple ? Simple(true) : new RleFilter(sized);
        return simple ? new RleFilter(sized) : Simple(false);
    };
    if (a.Runs.empty() && b.Runs.empty()) return Simple(calc(a.Default, b.Default));
    if (a.Runs.empty()) return withSimple(b, a.Default);
    if (b.Runs.empty()) return withSimple(a, b.Default);
    if (a.Count != b.Count) return nullptr;  // Y_ABORT_UNLESS (:453)
    size_t i = 0, j = 0;
    uint32_t pa = 0, pb = 0;
    bool va = a.Start(), vb = b.Start();
    RunSink sink;
    while (i < a.Runs.size() && j < b.Runs.size()) {
        uint32_t n = 0;
        if (!Overlap(pb, pb + b.Runs[j], pa, pa + a.Runs[i], n)) return nullptr;
        sink.Put(calc(va, vb), n);
        const uint32_t ea = pa + a.Runs[i], eb = pb + b.Runs[j];
        if (ea <= eb) {
            pa = ea;
            va = !va;
            ++i;
        }
        if (eb <= ea) {
            pb = eb;
            vb = !vb;
            ++j;
        }
    }
    if (i != a.Runs.size() || j != b.Runs.size()) return nullptr;  // Y_ABORT_UNLESS (:494)
    return sink.Finish();
}

// CombineSequentialAnd (arrow_filter.cpp:515-575)
RleFilter* CombineSequential(const RleFilter& self, const RleFilter& ext) {
    if (self.Runs.empty()) return self.Default ? new RleFilter(ext) : Simple(false);
    if (ext.Runs.empty()) return ext.Default ? new RleFilter(self) : Simple(false);
    size_t i = 0, j = 0;
    uint32_t ps = 0, pe = 0;  // both positions count rows of `ext`'s frame shifted by the rows `self` drops (:546-548)
    bool vs = self.Start(), ve = ext.Start();
    RunSink sink;
    while (i < self.Runs.size()) {
        if (vs && j >= ext.Runs.size()) return nullptr;  // Y_ABORT_UNLESS (:535)
        uint32_t n = self.Runs[i];
        if (vs && !Overlap(pe, pe + ext.Runs[j], ps, ps + self.Runs[i], n)) return nullptr;
        sink.Put(vs && ve, n);
        if (!vs) {
            pe += self.Runs[i];
            ps += self.Runs[i];
            vs = !vs;
            ++i;
            continue;
        }
        const uint32_t es = ps + self.Runs[i], ee = pe + ext.Runs[j];
        if (es <= ee) {
            ps = es;
            vs = !vs;
            ++i;
        }
        if (ee <= es) {
            pe = ee;
            ve = !ve;
            ++j;
        }
    }
    if (j != ext.Runs.size()) return nullptr;  // AFL_VERIFY (:568)
    return sink.Finish();
}

}  // namespace

extern "C" {

// TColumnFilter(std::vector<bool>&&) -> Reset(count, getter) (arrow_filter.h:132-155)
void* orc_filter_from_bools(const uint8_t* values, uint32_t n) {
    auto* f = new RleFilter();
    if (n == 0) return f;
    bool current = values[0] != 0;
    uint32_t same = 1;
    for (uint32_t i = 1; i < n; ++i) {
        if ((values[i] != 0) != current) {
            f->Add(current, same);
            same = 0;
            current = !current;
        }
        ++same;
    }
    f->Add(current, same);
    return f;
}

void* orc_filter_simple(int allow) { return Simple(allow != 0); }

void* orc_filter_merge(const void* a, const void* b, int isOr) {
    return Merge(*static_cast<const RleFilter*>(a), *static_cast<const RleFilter*>(b), isOr != 0);
}

void* orc_filter_combine_sequential_and(const void* self, const void* ext) {
    return CombineSequential(*static_cast<const RleFilter*>(self), *static_cast<const RleFilter*>(ext));
}

void orc_filter_info(const void* p, int* empty, int* defaultValue, int* lastValue, uint32_t* count, uint64_t* nruns) {
    const auto& f = *static_cast<const RleFilter*>(p);
    *empty = f.Runs.empty();
    *defaultValue = f.Default;
    *lastValue = f.Last;
    *count = f.Count;
    *nruns = f.Runs.size();
}

void orc_filter_runs(const void* p, uint32_t* out) {
    const auto& f = *static_cast<const RleFilter*>(p);
    std::copy(f.Runs.begin(), f.Runs.end(), out);
}

// BuildSimpleFilter (arrow_filter.cpp:383-400): one byte per row
void orc_filter_bools(const void* p, uint8_t* out) {
    const auto& f = *static_cast<const RleFilter*>(p);
    bool v = f.Start();
    uint32_t pos = 0;
    for (uint32_t n : f.Runs) {
        std::fill(out + pos, out + pos + n, uint8_t(v));
        pos += n;
        v = !v;
    }
}

// GetFilteredCount (arrow_filter.cpp:586-608); -1 for the empty optional
int64_t orc_filter_filtered_count(const void* p) {
    const auto& f = *static_cast<const RleFilter*>(p);
    if (f.TotalAllow()) return f.Count ? int64_t(f.Count) : -1;
    if (f.TotalDeny()) return 0;
    int64_t kept = 0;
    bool v = f.Start();
    for (uint32_t n : f.Runs) {
        if (v) kept += n;
        v = !v;
    }
    return kept;
}

void orc_filter_free(void* p) { delete static_cast<RleFilter*>(p); }

}  // extern "C"
