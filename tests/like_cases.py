"""SQL LIKE patterns beyond the three forms Arrow rewrites to plain searches: shared by the CPU and GPU tests.

`reference_regex` is the regular expression the reference's Arrow builds for such a pattern (MakeLikeRegex,
contrib/libs/apache/arrow/cpp/src/arrow/compute/kernels/scalar_string.cc:739-778), evaluated here with Python's `re` on
valid UTF-8 text (where `.` under DOTALL is one code point, as in RE2's UTF-8 mode) -- an independent engine fed the
reference's own translation.  The reference's tests hold no known answers for these patterns (parity unpinned)."""
import re

import numpy as np

PATTERNS = ["a_c", "_", "__", "%a%b%", "a%b", "%a_", "_%_", "ab\\%c", "ab\\_c%", "%\\\\%", "a\\", "", "%", "%%", "a%%b_c",
            "%ab%cd", "x%y%z", "_b%", "%é_", "日_%", "%_本%", "a.c", "a*c%x", "(%)", "[_]", "%^$%_"]


def reference_regex(pattern: str) -> str:
    out, escaped = "(?s:^", False
    for c in pattern:
        if not escaped and c == "%":
            out += ".*"
        elif not escaped and c == "_":
            out += "."
        elif not escaped and c == "\\":
            escaped = True
        else:
            out += ("\\" + c) if c in ".?+*^$\\[{()|" else c
            escaped = False
    return out + "$)"


def strings(rng: np.random.Generator, n: int):
    alphabet = list("abcxyz%_\\.*()[]^$") + ["é", "日", "本", "\n", " "]
    return ["".join(rng.choice(alphabet, rng.integers(0, 7))) for _ in range(n)]


def expected(values, pattern: str):
    # (Python's `$` also matches before a final newline, RE2's only at the very end: `\Z` is RE2's `$` here)
    rx = re.compile(reference_regex(pattern)[:-2] + "\\Z)")
    return [rx.search(v) is not None for v in values]
