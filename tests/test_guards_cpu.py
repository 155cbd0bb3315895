"""Guards that run in the GPU-less build container (verdict r3 item 6).
  * the CPU oracle under AddressSanitizer + UBSan (sanitizers run on the CPU build only);
  * the compiler's resource report for the kernels that sit on the register allocator's edge: a compiler or source change that spills
    inside the row loop of the headline kernel costs 0.5 ms per 1080p launch and nothing else would notice it here."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gcc_file(name):
    return subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()


def test_oracle_suite_under_address_sanitizer():
    """`make -C oracle asan` and tests/test_oracle_cpu.py against libdfe_oracle_asan.so (loaded through DFE_ORACLE_SO, the sanitizer
    runtimes preloaded): any out-of-bounds access, use after free or undefined behaviour in the oracle aborts the child."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "libdfe_oracle_asan.so")
    asan, ubsan = _gcc_file("libasan.so"), _gcc_file("libubsan.so")
    if not (os.path.isabs(asan) and os.path.exists(asan)):
        pytest.skip("no libasan in this toolchain")
    env = dict(os.environ, DFE_ORACLE_SO=so, LD_PRELOAD=asan + (":" + ubsan if os.path.isabs(ubsan) else ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_cpu.py"), os.path.join(ROOT, "tests", "test_egomotion_cpu.py"),
                        "-q", "-x", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:] + r.stderr[-3000:])
    assert "passed" in r.stdout and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
    # the child really ran on the sanitizer build
    probe = subprocess.run([sys.executable, "-c", "from tests import oracle as o; o.lib(); print(any('libdfe_oracle_asan' in l for l in open('/proc/self/maps')))"],
                           env=env, capture_output=True, text=True, cwd=ROOT)
    assert probe.stdout.strip().endswith("True"), probe.stdout + probe.stderr


def _kres(src, substr, *defs):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kres.py"), os.path.join(ROOT, "depth-estimation_amd", "csrc", src), substr, *defs],
                         capture_output=True, text=True).stdout
    rows = {}
    for line in out.splitlines():
        m = re.match(r"(.*?)\s+VGPR (\d+) scratch (\d+) sgpr-spill (\d+)", line)
        if m:
            rows[m.group(1).strip()] = tuple(int(x) for x in m.groups()[1:])
    return rows


def test_headline_kernels_stay_inside_the_register_file():
    """tools/kres.py (the compiler's -Rpass-analysis=kernel-resource-usage remarks):
      * the fused 3-channel sweep ssd_cv_rowimg_kernel<3,7,8,true,true,true,1089,false> -- the bench's dominant kernel -- at <= 128 VGPRs,
        <= 8 B of scratch (the two spills outside the row loop) and <= 40 spilled SGPRs (39 at cv-r4.1);
      * the plain 33 x 33 build <3,7,8,true,false,true,1089,false> at <= 8 B of scratch (round 5: the thread id parked at kernel entry and
        reloaded once per piece, outside the row loop -- checked on the ISA; none before the scalar-carry change)."""
    rows = _kres("ssd_cost_volume.hip", "rowimg")
    fused = [v for k, v in rows.items() if k.endswith("ssd_cv_rowimg_kernel<3, 7, 8, true, true, true, 1089, false>")]
    plain = [v for k, v in rows.items() if k.endswith("ssd_cv_rowimg_kernel<3, 7, 8, true, false, true, 1089, false>")]
    assert len(fused) == 1 and len(plain) == 1, sorted(rows)
    vgpr, scratch, spill = fused[0]
    assert vgpr <= 128 and scratch <= 8 and spill <= 40, "fused sweep: %d VGPRs, %d B scratch, %d spilled SGPRs" % fused[0]
    assert plain[0][0] <= 128 and plain[0][1] <= 8, "plain build: %d VGPRs, %d B scratch" % plain[0][:2]


def test_flat_matcher_kernels_do_not_spill_in_the_plane_loop():
    """feat_matching_flat_kernel: the 16-wide and the 17-wide (no extra row) instantiations without scratch, the 17 x 17 one with its
    extra task at <= 16 B (three values parked around the plane loop, none inside it: checked on the ISA when it was written)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kres.py"), os.path.join(ROOT, "depth-estimation_amd", "csrc", "feat_matching_flat.hip"), "flat"],
                         capture_output=True, text=True).stdout
    vals = {m[0]: tuple(int(x) for x in m[1:]) for m in re.findall(r"feat_matching_flat_kernel<([^>]*)>\s+VGPR (\d+) scratch (\d+) sgpr-spill (\d+)", out)}
    assert len(vals) == 9, out          # <MW, EXTRA, MODE>: 16 / 17 wide, 17 wide with the extra row; volume (0), arg-min (1), soft-max (2)
    assert all(v[0] <= 128 for v in vals.values())
    assert vals["17, true, 1"][1] <= 40 and vals["17, true, 0"][1] <= 16 and vals["17, true, 2"][1] <= 16, vals
    assert all(v[1] == 0 for k, v in vals.items() if "true" not in k), vals
