// Per-element semantics of the scan-VM instructions, shared by the HIP kernels and by the
// plan compiler's constant folder so a scalar-scalar expression folds to exactly what the
// kernel would have computed row by row.
//
// Semantics restated from the reference's delegates:
//  * Arrow 5.0 arithmetic: unchecked wrap-around integers, int x/0 -> error, MIN/-1 -> 0
//    (contrib/libs/apache/arrow/cpp/src/arrow/compute/kernels/scalar_arithmetic.cc:122-327)
//  * Arrow compare on the common numeric type (scalar_compare.cc:36-66, codegen_internal.cc:226-278)
//  * YQL block kernels: value-correct mixed-sign compare, x/0 and MIN/-1 -> NULL
//    (ydb/library/yql/minikql/invoke_builtins/mkql_builtins_equals.cpp:13-44,
//     mkql_builtins_less.cpp:13-44, mkql_builtins_div.cpp:30-48)
#pragma once
#if !defined(__HIPCC_RTC__)
#include <math.h>
#endif
#include "vm.h"

namespace ssa {

SSA_HD double BitsToF64(uint64_t b) {
    double d;
    __builtin_memcpy(&d, &b, 8);
    return d;
}
SSA_HD uint64_t F64ToBits(double d) {
    uint64_t b;
    __builtin_memcpy(&b, &d, 8);
    return b;
}
SSA_HD float BitsToF32(uint32_t b) {
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}

// Wrap an integer payload to the width/signedness of integer type `t`.
SSA_HD uint64_t WrapTo(uint64_t v, int t) {
    switch (t) {
        case T_I8: return uint64_t(int64_t(int8_t(v)));
        case T_I16: return uint64_t(int64_t(int16_t(v)));
        case T_I32: return uint64_t(int64_t(int32_t(v)));
        case T_U8: return uint64_t(uint8_t(v));
        case T_U16: return uint64_t(uint16_t(v));
        case T_U32: return uint64_t(uint32_t(v));
        case T_BOOL: return v != 0 ? 1 : 0;
        default: return v;
    }
}

// double -> int64 / int32 as the reference's x86-64 build converts (cvttsd2si): NaN and everything outside the target's
// range give INT64_MIN / INT32_MIN.  (The conversion is undefined in C++; the GPU's own instruction saturates and turns
// NaN into 0.)
SSA_HD int64_t F64ToI64LikeX86(double d) {
    if (!(d >= -9223372036854775808.0 && d < 9223372036854775808.0)) return INT64_MIN;
    return int64_t(d);
}
SSA_HD int32_t F64ToI32LikeX86(double d) {
    if (!(d > -2147483649.0 && d < 2147483648.0)) return (-2147483647 - 1);
    return int32_t(d);
}

// static_cast between numeric representations.  `from` is the payload class, `to` a TypeId.  A floating value goes to
// an integer type as the reference's build (clang for x86-64 without AVX-512) compiles static_cast<T>(double), also
// where C++ leaves the result open: int64 by cvttsd2si, targets of up to 32 bits through int32 (uint32 through int64)
// and then truncated, uint64 as cvttsd2si(x) | (cvttsd2si(x - 2^63) & sign(cvttsd2si(x))).
SSA_HD uint64_t CastVal(uint64_t v, int from, int to) {
    const bool toF = (to == T_F32 || to == T_F64);
    if (from == CLS_I || from == CLS_U) {
        if (to == T_BOOL) return v != 0 ? 1 : 0;
        if (!toF) return WrapTo(v, to);
        double d = (from == CLS_I) ? double(int64_t(v)) : double(v);
        if (to == T_F32) {
            float f = (from == CLS_I) ? float(int64_t(v)) : float(v);
            d = double(f);
        }
        return F64ToBits(d);
    }
    // floating source (F32 payloads are exact doubles)
    double d = BitsToF64(v);
    if (to == T_F64) return v;
    if (to == T_F32) return F64ToBits(double(float(d)));
    if (to == T_BOOL) return d != 0.0 ? 1 : 0;
    switch (to) {
        case T_I64: case T_TS: case T_DUR:
            return uint64_t(F64ToI64LikeX86(d));
        case T_U64: {
            const int64_t a = F64ToI64LikeX86(d), b = F64ToI64LikeX86(d - 9223372036854775808.0);
            return uint64_t(a) | (uint64_t(b) & uint64_t(a >> 63));
        }
        case T_U32:
            return WrapTo(uint64_t(F64ToI64LikeX86(d)), to);
        default:
            return WrapTo(uint64_t(int64_t(F64ToI32LikeX86(d))), to);
    }
}

// Would a *safe* cast of this value to `to` succeed?  (Arrow's implicit casts inserted by
// DispatchBest are CastOptions::Safe: compute/cast.cc:256-270.)
SSA_HD bool FitsType(uint64_t v, int from, int to) {
    if (from == CLS_I) {
        int64_t s = int64_t(v);
        switch (to) {
            case T_I8: return s >= -128 && s <= 127;
            case T_I16: return s >= -32768 && s <= 32767;
            case T_I32: return s >= -2147483648LL && s <= 2147483647LL;
            case T_U8: return s >= 0 && s <= 255;
            case T_U16: return s >= 0 && s <= 65535;
            case T_U32: return s >= 0 && s <= 4294967295LL;
            case T_U64: return s >= 0;
            case T_F32: return s >= -(1LL << 24) && s <= (1LL << 24);
            case T_F64: return s >= -(1LL << 53) && s <= (1LL << 53);
            default: return true;
        }
    }
    if (from == CLS_U) {
        switch (to) {
            case T_I8: return v <= 127;
            case T_I16: return v <= 32767;
            case T_I32: return v <= 2147483647ULL;
            case T_I64: return v <= 9223372036854775807ULL;
            case T_U8: return v <= 255;
            case T_U16: return v <= 65535;
            case T_U32: return v <= 4294967295ULL;
            case T_F32: return v <= (1ULL << 24);
            case T_F64: return v <= (1ULL << 53);
            default: return true;
        }
    }
    return true;
}

struct ArithResult {
    uint64_t v;
    bool null;   // YQL: result is NULL
    bool error;  // Arrow: divide by zero
    bool overflow;  // Arrow checked arithmetic: the result does not fit the type
};

// ADD/SUB/MUL on payloads; integer results still need WrapTo(result type).
SSA_HD uint64_t ArithBasic(int op, int cls, uint64_t a, uint64_t b) {
    if (cls == CLS_F64) {
        double x = BitsToF64(a), y = BitsToF64(b);
        double r = op == OP_ADD ? x + y : op == OP_SUB ? x - y : x * y;
        return F64ToBits(r);
    }
    if (cls == CLS_F32) {
        float x = float(BitsToF64(a)), y = float(BitsToF64(b));
        float r = op == OP_ADD ? x + y : op == OP_SUB ? x - y : x * y;
        return F64ToBits(double(r));
    }
    return op == OP_ADD ? a + b : op == OP_SUB ? a - b : a * b;
}

// `t` is the result integer type: INT_MIN of *that* width divided by -1 is the overflow case.
SSA_HD ArithResult ArithDiv(bool mod, int cls, int dialect, int t, uint64_t a, uint64_t b) {
    ArithResult r{0, false, false};
    if (cls == CLS_F64) {
        double x = BitsToF64(a), y = BitsToF64(b);
        r.v = F64ToBits(mod ? __builtin_fmod(x, y) : x / y);
        return r;
    }
    if (cls == CLS_F32) {
        float x = float(BitsToF64(a)), y = float(BitsToF64(b));
        r.v = F64ToBits(double(mod ? __builtin_fmodf(x, y) : x / y));
        return r;
    }
    if (b == 0) {
        if (dialect == DIALECT_YQL) r.null = true; else r.error = true;
        return r;
    }
    if (cls == CLS_U) {
        r.v = mod ? a % b : a / b;
        return r;
    }
    int64_t x = int64_t(a), y = int64_t(b);
    int64_t lo = (t == T_I8) ? -128 : (t == T_I16) ? -32768 : (t == T_I32) ? -2147483648LL : INT64_MIN;
    if (y == -1 && x == lo) {
        if (dialect == DIALECT_YQL) r.null = true;  // mkql_builtins_div.cpp / mod: MIN / -1 -> NULL
        r.v = 0;                                   // Arrow: result = 0
        return r;
    }
    r.v = uint64_t(mod ? x % y : x / y);
    return r;
}

// The libm functions of ydb/library/arrow_kernels/func_math.h:7-170: a float argument is computed in float (std::acosh(float)
// is the float overload) and widened, everything else in double.  The device's math library and the host's differ in the
// last bits of these functions.
SSA_HD double MathFunc(int fn, bool f32, double x, double y) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SSA_SPEC)
    // The generic interpreter leaves these out (their code costs every program registers); the executor builds a
    // kernel for any program that uses one and refuses to run it otherwise.
    (void)fn; (void)f32; (void)x; (void)y;
    return 0;
#else
    if (f32) {
        const float p = float(x), q = float(y);
        switch (fn) {
            case FN_ACOSH: return double(::acoshf(p));
            case FN_ATANH: return double(::atanhf(p));
            case FN_CBRT: return double(::cbrtf(p));
            case FN_COSH: return double(::coshf(p));
            case FN_ERF: return double(::erff(p));
            case FN_ERFC: return double(::erfcf(p));
            case FN_EXP: return double(::expf(p));
            case FN_EXP2: return double(::exp2f(p));
            case FN_LGAMMA: return double(::lgammaf(p));
            case FN_SINH: return double(::sinhf(p));
            case FN_TGAMMA: return double(::tgammaf(p));
            case FN_HYPOT: return double(::hypotf(p, q));
            default: return 0;
        }
    }
    switch (fn) {
        case FN_ACOSH: return ::acosh(x);
        case FN_ATANH: return ::atanh(x);
        case FN_CBRT: return ::cbrt(x);
        case FN_COSH: return ::cosh(x);
        case FN_ERF: return ::erf(x);
        case FN_ERFC: return ::erfc(x);
        case FN_EXP: return ::exp(x);
        case FN_EXP2: return ::exp2(x);
        case FN_LGAMMA: return ::lgamma(x);
        case FN_SINH: return ::sinh(x);
        case FN_TGAMMA: return ::tgamma(x);
        case FN_HYPOT: return ::hypot(x, y);
        default: return 0;
    }
#endif
}

// x % y as C++ computes it in the operands' type; MIN % -1 (undefined there) is 0.
SSA_HD uint64_t IntRem(bool sgn, uint64_t a, uint64_t b) {
    if (!sgn) return a % b;
    const int64_t x = int64_t(a), y = int64_t(b);
    return y == -1 ? 0 : uint64_t(x % y);
}

// One element of an OP_FUNC function on payloads of type `t` (both operands were cast to it).
//  * gcd / lcm (ydb/library/arrow_kernels/func_gcd.h:20-36, func_lcm.h:6-22): Euclid with the C++ remainder, so the
//    sign follows the operands; lcm = (lhs * rhs wrapped to the type) / gcd, gcd 0 -> "divide by zero".
//  * mod / modOrZero (func_modulo.h:6-30, func_modulo_or_zero.h:6-28): floats are converted to int64 first.
//  * round / roundBankers / roundToExp2 (func_round.h:23-76): integers go through double and back; roundToExp2 is the
//    largest power of two not above the argument (0 for arguments <= 0), for floats the value with its mantissa cleared.
//  * abs / negate (compute/kernels/scalar_arithmetic.cc:81-96,362-378): wrap-around integers.
//  * floor / ceil / trunc (scalar_arithmetic.cc:820-838) and sqrt (func_math.h:152-160) see floating operands only.
// The kernels are applied to every slot, null ones included (ScalarBinary / ScalarUnary applicators), so `error` does
// not depend on validity.  Float payloads of T_F32 operands are exact doubles; results are rounded to float again.
SSA_HD ArithResult ApplyFunc(int fn, int t, uint64_t a, uint64_t b) {
    ArithResult r{0, false, false, false};
    const bool flt = (t == T_F32 || t == T_F64);
    const bool sgn = IsSignedInt(t) || IsTemporal(t);
    switch (fn) {
        case FN_GCD: case FN_LCM: {
            uint64_t x = a, y = b;
            while (y != 0) {
                const uint64_t m = IntRem(sgn, x, y);
                x = y;
                y = m;
            }
            if (fn == FN_GCD) {
                r.v = x;
                break;
            }
            if (x == 0) {
                r.error = true;
                break;
            }
            const uint64_t prod = WrapTo(a * b, t);
            if (!sgn) r.v = prod / x;
            else r.v = int64_t(x) == -1 ? uint64_t(0) - prod : uint64_t(int64_t(prod) / int64_t(x));
            r.v = WrapTo(r.v, t);
            break;
        }
        case FN_MOD: case FN_MODZ: {
            uint64_t x = a, y = b;
            bool s = sgn;
            if (flt) {
                x = uint64_t(F64ToI64LikeX86(BitsToF64(a)));
                y = uint64_t(F64ToI64LikeX86(BitsToF64(b)));
                s = true;
            }
            if (y == 0) {
                r.error = (fn == FN_MOD);
                r.v = flt ? F64ToBits(0.0) : 0;
                break;
            }
            const uint64_t m = IntRem(s, x, y);
            if (t == T_F64) r.v = F64ToBits(double(int64_t(m)));
            else if (t == T_F32) r.v = F64ToBits(double(float(int64_t(m))));
            else r.v = m;
            break;
        }
        case FN_ABS:
            if (flt) r.v = F64ToBits(__builtin_fabs(BitsToF64(a)));
            else r.v = (sgn && int64_t(a) < 0) ? WrapTo(uint64_t(0) - a, t) : a;
            break;
        case FN_NEG:
            if (flt) r.v = F64ToBits(-BitsToF64(a));
            else r.v = WrapTo(uint64_t(0) - a, t);
            break;
        case FN_ROUND: case FN_ROUNDBANKERS: {
            double d = flt ? BitsToF64(a) : sgn ? double(int64_t(a)) : double(a);
            d = fn == FN_ROUND ? __builtin_round(d) : __builtin_rint(d);
            if (flt) r.v = F64ToBits(d);
            else if (sgn) r.v = WrapTo(d >= 9223372036854775808.0 ? uint64_t(INT64_MIN) : uint64_t(int64_t(d)), t);
            else r.v = WrapTo(d >= 18446744073709551616.0 ? 0 : uint64_t(d), t);
            break;
        }
        case FN_ROUNDEXP2:
            if (t == T_F64) {
                r.v = a & ~((uint64_t(1) << 52) - 1);
            } else if (t == T_F32) {
                const float f = float(BitsToF64(a));
                uint32_t bits;
                __builtin_memcpy(&bits, &f, 4);
                bits &= ~((uint32_t(1) << 23) - 1);
                r.v = F64ToBits(double(BitsToF32(bits)));
            } else if (a == 0 || (sgn && int64_t(a) < 0)) {
                r.v = 0;
            } else {
                r.v = uint64_t(1) << (63 - __builtin_clzll(a));
            }
            break;
        case FN_FLOOR: r.v = F64ToBits(__builtin_floor(BitsToF64(a))); break;
        case FN_CEIL: r.v = F64ToBits(__builtin_ceil(BitsToF64(a))); break;
        case FN_TRUNC: r.v = F64ToBits(__builtin_trunc(BitsToF64(a))); break;
        case FN_SQRT:
            if (t == T_F32) r.v = F64ToBits(double(__builtin_sqrtf(float(BitsToF64(a)))));
            else r.v = F64ToBits(__builtin_sqrt(BitsToF64(a)));
            break;
        case FN_ADD_CHECKED: case FN_SUB_CHECKED: case FN_MUL_CHECKED: case FN_DIV_CHECKED: {
            // add_checked ... divide_checked (compute/kernels/scalar_arithmetic.cc:147-169,329-360)
            if (flt) {
                if (fn == FN_DIV_CHECKED && BitsToF64(b) == 0.0) {
                    r.error = true;
                    r.v = F64ToBits(0.0);
                    break;
                }
                const int cls = t == T_F32 ? CLS_F32 : CLS_F64;
                if (fn == FN_DIV_CHECKED) r.v = ArithDiv(false, cls, DIALECT_ARROW, t, a, b).v;
                else r.v = ArithBasic(fn == FN_ADD_CHECKED ? OP_ADD : fn == FN_SUB_CHECKED ? OP_SUB : OP_MUL, cls, a, b);
                break;
            }
            if (fn == FN_DIV_CHECKED) {
                if (b == 0) {
                    r.error = true;
                    break;
                }
                if (sgn) {
                    const int64_t lo = (t == T_I8) ? -128 : (t == T_I16) ? -32768 : (t == T_I32) ? -2147483648LL : INT64_MIN;
                    if (int64_t(a) == lo && int64_t(b) == -1) r.overflow = true;
                    else r.v = uint64_t(int64_t(a) / int64_t(b));
                } else {
                    r.v = a / b;
                }
                break;
            }
            bool over;
            if (sgn) {
                long long v = 0;
                const long long x = (long long)int64_t(a), y = (long long)int64_t(b);
                over = fn == FN_ADD_CHECKED ? __builtin_saddll_overflow(x, y, &v)
                       : fn == FN_SUB_CHECKED ? __builtin_ssubll_overflow(x, y, &v) : __builtin_smulll_overflow(x, y, &v);
                r.v = uint64_t(v);
            } else {
                unsigned long long v = 0;
                over = fn == FN_ADD_CHECKED ? __builtin_uaddll_overflow(a, b, &v)
                       : fn == FN_SUB_CHECKED ? __builtin_usubll_overflow(a, b, &v) : __builtin_umulll_overflow(a, b, &v);
                r.v = uint64_t(v);
            }
            if (!over && WrapTo(r.v, t) != r.v) over = true;  // fits 64 bits but not the column's type
            r.overflow = over;
            break;
        }
        default:
            r.v = F64ToBits(MathFunc(fn, t == T_F32, BitsToF64(a), BitsToF64(b)));
            break;
    }
    return r;
}

SSA_HD bool CompareVals(int cmp, int cls, uint64_t a, uint64_t b) {
    int lt, eq;  // a < b, a == b (unordered floats: both 0 and ne = 1)
    bool unordered = false;
    if (cls == CLS_F64 || cls == CLS_F32) {
        double x = BitsToF64(a), y = BitsToF64(b);
        lt = x < y;
        eq = x == y;
        unordered = (x != x) || (y != y);
    } else if (cls == CLS_I) {
        lt = int64_t(a) < int64_t(b);
        eq = a == b;
    } else if (cls == CLS_U) {
        lt = a < b;
        eq = a == b;
    } else if (cls == CLS_IU) {  // a signed, b unsigned
        bool neg = int64_t(a) < 0;
        lt = neg || a < b;
        eq = !neg && a == b;
    } else {  // CLS_UI: a unsigned, b signed
        bool neg = int64_t(b) < 0;
        lt = !neg && a < b;
        eq = !neg && a == b;
    }
    switch (cmp) {
        case CMP_EQ: return eq;
        case CMP_NE: return !eq;
        case CMP_LT: return lt;
        case CMP_LE: return lt || eq;
        case CMP_GT: return !unordered && !lt && !eq;
        default: return !unordered && !lt;  // CMP_GE
    }
}

SSA_HD uint8_t AsciiLower(uint8_t c) { return (c >= 'A' && c <= 'Z') ? uint8_t(c + 32) : c; }

// Length of the UTF-8 sequence that starts at s (n bytes left), 0 when the bytes there are not one: the sequences
// RE2's `.` accepts in its UTF-8 mode (shortest forms only, up to U+10FFFF).
SSA_HD int32_t Utf8CharLen(const uint8_t* s, int32_t n) {
    if (n <= 0) return 0;
    const uint8_t c = s[0];
    if (c < 0x80) return 1;
    auto cont = [](uint8_t b) { return (b & 0xC0) == 0x80; };
    if (c >= 0xC2 && c <= 0xDF) return (n >= 2 && cont(s[1])) ? 2 : 0;
    if (c >= 0xE0 && c <= 0xEF) {
        if (n < 3 || !cont(s[1]) || !cont(s[2])) return 0;
        return (c == 0xE0 && s[1] < 0xA0) ? 0 : 3;
    }
    if (c >= 0xF0 && c <= 0xF4) {
        if (n < 4 || !cont(s[1]) || !cont(s[2]) || !cont(s[3])) return 0;
        if (c == 0xF0 && s[1] < 0x90) return 0;
        if (c == 0xF4 && s[1] > 0x8F) return 0;
        return 4;
    }
    return 0;
}

// SQL LIKE as Arrow 5 evaluates a pattern it cannot turn into a plain search (compute/kernels/scalar_string.cc:739-778,
// MakeLikeRegex): the whole string must match; an unescaped '%' is any run of characters, an unescaped '_' one
// character (a UTF-8 sequence: the regular expression runs in UTF-8 mode with "dot matches newline"), a backslash
// makes the next byte literal and a trailing lone backslash stands for nothing.  One '%' is remembered to come back to;
// it gives up one more character on every mismatch.
SSA_HD bool LikeMatch(const uint8_t* s, int32_t n, const uint8_t* p, int32_t m) {
    int32_t si = 0, pi = 0, starP = -1, starS = 0;
    for (;;) {
        bool ok = false;
        if (pi < m) {
            const uint8_t c = p[pi];
            if (c == '%') {
                starP = ++pi;
                starS = si;
                continue;
            }
            if (c == '\\' && pi + 1 >= m) {  // lone backslash at the end
                ++pi;
                continue;
            }
            if (c == '_') {
                const int32_t l = Utf8CharLen(s + si, n - si);
                if (l > 0) {
                    si += l;
                    ++pi;
                    ok = true;
                }
            } else {
                const bool escaped = c == '\\';
                const uint8_t lit = escaped ? p[pi + 1] : c;
                if (si < n && s[si] == lit) {
                    ++si;
                    pi += escaped ? 2 : 1;
                    ok = true;
                }
            }
        } else if (si == n) {
            return true;
        }
        if (ok) continue;
        if (starP < 0) return false;
        const int32_t l = Utf8CharLen(s + starS, n - starS);
        if (l == 0) return false;  // nothing (or no character) left for the '%' to take
        starS += l;
        si = starS;
        pi = starP;
    }
}

// Byte-wise substring / prefix / suffix test (std::string_view semantics; an empty pattern
// matches everything).  ignore_case folds ASCII letters only.
SSA_HD bool StrMatch(int kind, bool icase, const uint8_t* s, int32_t n, const uint8_t* p, int32_t m) {
    if (kind == STR_LIKE) return LikeMatch(s, n, p, m);
    if (m > n) return false;
    if (m == 0) return true;
    int32_t first = 0, last = 0;
    if (kind == STR_STARTS) {
        first = last = 0;
    } else if (kind == STR_ENDS) {
        first = last = n - m;
    } else {
        first = 0;
        last = n - m;
    }
    for (int32_t i = first; i <= last; ++i) {
        int32_t j = 0;
        if (icase) {
            while (j < m && AsciiLower(s[i + j]) == AsciiLower(p[j])) ++j;
        } else {
            while (j < m && s[i + j] == p[j]) ++j;
        }
        if (j == m) return true;
    }
    return false;
}

// memcmp-style ordering of two byte strings (std::string_view operator<=>).
SSA_HD int StrCompare(const uint8_t* a, int32_t n, const uint8_t* b, int32_t m) {
    int32_t k = n < m ? n : m;
    for (int32_t i = 0; i < k; ++i) {
        if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    }
    return n < m ? -1 : (n > m ? 1 : 0);
}

SSA_HD bool CmpFromOrder(int cmp, int ord) {
    switch (cmp) {
        case CMP_EQ: return ord == 0;
        case CMP_NE: return ord != 0;
        case CMP_LT: return ord < 0;
        case CMP_LE: return ord <= 0;
        case CMP_GT: return ord > 0;
        default: return ord >= 0;
    }
}

}  // namespace ssa
