"""The loss kernels' direction-bin functions (csrc/edge_direction.hpp: four integer compares + a table; three range tests + the sign) against the literal restatement of
GradLayer's masks (reference packnet_sfm/losses/grad_loss.py:80-93) kept in the same header: compiled for the host with g++ and compared
over float bit patterns -- every pattern within 4096 ulps of +-k*pi/8, the special values, and every 257th pattern of all 2^32 (set
MTE_EXHAUSTIVE=1 for all of them: ~25 s)."""
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HARNESS = r"""
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "edge_direction.hpp"
static long bad = 0, n = 0;
static void check(uint32_t v) {
    float f; std::memcpy(&f, &v, 4);
    const int a = direction_code_literal(f), b = direction_code(f), c = code_from_masks(direction_masks(f));
    if ((a != b || a != c) && bad++ < 10) std::printf("mismatch %08x %g literal %d fast %d masks %d\n", v, f, a, b, c);
    ++n;
}
int main(int argc, char** argv) {
    const uint64_t stride = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 257;
    for (uint64_t u = 0; u < (1ull << 32); u += stride) check((uint32_t)u);
    const double pi = 3.14159265358979323846;
    for (int k = 1; k <= 7; k += 2)
        for (int sgn = 0; sgn < 2; ++sgn) {
            const float t = (float)(k * pi / 8);
            uint32_t c; std::memcpy(&c, &t, 4);
            c |= sgn ? 0x80000000u : 0u;
            for (int d = -4096; d <= 4096; ++d) check(c + (uint32_t)d);
        }
    const uint32_t special[] = {0u, 0x80000000u, 1u, 0x80000001u, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0xffc00000u, 0x7f7fffffu, 0xff7fffffu, 0x40490fdbu, 0xc0490fdbu};
    for (uint32_t v : special) check(v);
    // one representative per bin: the literal masks themselves must give the documented codes
    const float mid[] = {0.f, 0.785f, 1.571f, 2.356f, 3.1f, -0.785f, -1.571f, -2.356f, -3.1f};
    const int want[] = {0, 2, 1, 3, 0, 3, 1, 2, 0};
    for (int i = 0; i < 9; ++i) if (direction_code(mid[i]) != want[i]) { std::printf("bin %g: %d, expected %d\n", mid[i], direction_code(mid[i]), want[i]); ++bad; }
    std::printf("checked %ld patterns, %ld mismatches\n", n, bad);
    return bad != 0;
}
"""


def test_fast_direction_code_equals_the_literal_masks(tmp_path):
    src = tmp_path / "dc.cpp"
    src.write_text(HARNESS)
    exe = tmp_path / "dc"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "mindtheedge_amd", "csrc"), "-o", str(exe), str(src)])
    stride = "1" if os.environ.get("MTE_EXHAUSTIVE") else "257"
    out = subprocess.run([str(exe), stride], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 mismatches" in out.stdout


def test_fast_direction_code_matches_the_oracle(tmp_path):
    """the same function against the oracle's own direction selection (oracle/loss_oracle.py::direction_code, torch) on random angles and on the bin edges"""
    import torch
    from oracle import loss_oracle
    rng = np.random.default_rng(5)
    edges = np.array([s * k * np.pi / 8 for k in range(0, 9) for s in (-1.0, 1.0)], dtype=np.float32)
    ang = np.concatenate([rng.uniform(-3.3, 3.3, 20000).astype(np.float32), edges, np.nextafter(edges, np.float32(10)), np.nextafter(edges, np.float32(-10))])
    want = loss_oracle.direction_code(torch.from_numpy(ang)).numpy()
    src = tmp_path / "dc2.cpp"
    src.write_text('#include <cstdio>\n#include "edge_direction.hpp"\nint main() { float f; while (std::fread(&f, 4, 1, stdin) == 1) std::putchar(\'0\' + direction_code(f)); return 0; }\n')
    exe = tmp_path / "dc2"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "mindtheedge_amd", "csrc"), "-o", str(exe), str(src)])
    out = subprocess.run([str(exe)], input=ang.tobytes(), capture_output=True)
    got = np.frombuffer(out.stdout, dtype=np.uint8).astype(np.int64) - ord("0")
    assert got.shape == want.shape and np.array_equal(got, want)
