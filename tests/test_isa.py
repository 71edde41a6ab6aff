"""CPU (hipcc cross-compiles): properties of the GENERATED gfx950 code of the scan kernels that the C++ source
cannot express and a compiler bump could silently break.

1. The part of the look-up table that is gathered through L1 lives in global memory and is written by all
   wavefronts of a workgroup before a barrier: every wavefront must release its stores BEFORE the barrier
   (s_waitcnt vmcnt(0) between its last table store and the barrier) and invalidate its L1 AFTER it (buffer_inv)
   -- VERDICT r1's latent issue.
2. The pre-filter's hot loop must not touch scratch memory: a spilled queue pointer cost 10 %, spilled code
   words 8 % + 4 GB of writes per launch in round 2."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def scan_asm(tmp_path_factory):
    if not os.path.isfile(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "rq_scan.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S",
                           "--cuda-device-only", os.path.join(ROOT, "rayuela.jl_amd", "csrc", "rq_scan.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernel(asm, m, filt):
    name = "_ZN2rq15adc_scan_kernelILi%dELb0ELb%dELb0EEEvNS_10ScanParamsE" % (m, 1 if filt else 0)
    start = asm.index("\n" + name + ":")
    end = asm.index("s_endpgm", start)
    return [ln for ln in asm[start:end].splitlines() if ln.strip() and not ln.lstrip().startswith(";")]


@pytest.mark.parametrize("m,filt", [(8, True), (8, False), (16, True), (4, False), (32, False)])
def test_l1_table_release_acquire_around_the_barrier(scan_asm, m, filt):
    body = _kernel(scan_asm, m, filt)
    inv = [i for i, ln in enumerate(body) if "buffer_inv" in ln]
    assert inv, "no buffer_inv: the acquire fence after the table build is gone"
    i = inv[0]
    b = max(j for j in range(i) if "s_barrier" in body[j])
    assert i - b <= 3, "the acquire (buffer_inv) does not follow the table-build barrier"
    # release side: every global store issued before the barrier has been waited for
    st = max(j for j in range(b) if re.search(r"\bglobal_store", body[j]))
    between = body[st + 1:b]
    assert any(re.search(r"s_waitcnt[^\n]*vmcnt\(0\)", ln) for ln in between), \
        "no s_waitcnt vmcnt(0) between the last table store and the barrier:\n" + "\n".join(body[b - 8:b + 2])


@pytest.mark.parametrize("m,fine", [(8, 0), (8, 1), (16, 0)])
def test_prefilter_hot_loop_has_no_scratch_traffic(scan_asm, m, fine):
    name = "_ZN2rq15adc_scan_kernelILi%dELb0ELb1ELb%dEEEvNS_10ScanParamsE" % (m, fine)
    start = scan_asm.index("\n" + name + ":")
    lines = scan_asm[start:scan_asm.index("s_endpgm", start)].splitlines()
    lo = [i for i, ln in enumerate(lines) if "RQ_FILTER_LOOP_BEGIN" in ln]
    mid = [i for i, ln in enumerate(lines) if "RQ_FILTER_GATHERS_END" in ln]
    hi = [i for i, ln in enumerate(lines) if "RQ_FILTER_LOOP_END" in ln]
    # one marker each: BEGIN .. GATHERS_END is the streaming part (gathers, byte sums, alive bits), GATHERS_END .. LOOP_END the
    # queueing of the alive rows with the call sites of the exact re-evaluation
    assert len(lo) == 1 and len(mid) == 1 and len(hi) == 1 and lo[0] < mid[0] < hi[0], (lo, mid, hi)
    stream = lines[lo[0]:mid[0]]
    gathers = sum(1 for ln in stream if re.search(r"\bds_read_b(64|32)\b", ln))
    assert gathers >= 64, gathers                     # 8 rows x 8 (m = 8) / 4 rows x 16 (m = 16) gathers per block
    assert not [ln for ln in stream if "scratch_" in ln], "scratch access in the streaming part of the pre-filter loop"
    assert not [ln for ln in stream if "s_swappc" in ln], "a call inside the streaming part of the pre-filter loop"
    region = lines[lo[0]:hi[0]]
    stores = [ln for ln in region if "scratch_store" in ln]
    loads = [ln for ln in region if "scratch_load" in ln]
    # Reloads right after a call of the exact re-evaluation belong to that (rare) path; the streaming path itself --
    # everything that is not within a few instructions after an s_swappc -- must not touch scratch at all.
    hot = []
    since_call = 99
    for ln in region:
        if "s_swappc" in ln:
            since_call = 0
        elif ln.startswith(".LBB"):
            since_call = 99          # a label: reachable without the call
        elif ln.strip() and not ln.lstrip().startswith((";", ".")):
            since_call += 1
        if "scratch_" in ln and since_call > 12:
            hot.append(ln)
    assert not hot, "scratch access on the streaming path of the pre-filter loop:\n" + "\n".join(hot[:8])


# ---- encode_pq_split_kernel: inline asm never reads MFMA results the compiler has not waited for ------------------------
# The hazard recogniser does not look inside inline asm: an asm compare issued straight after an MFMA read stale accumulator
# registers now and then in round 3 (2-12 wrong codes per 1e7, csrc/rq_encode.hip near the tile re-run).  The asm that is
# left (the exec-masked v_mov_b64 copy of the winning tile, mask_leq16's v_cmp / v_addc) is only safe behind a
# COMPILER-VISIBLE VALU read of the same MFMA's destination registers (tile_min's v_min3 tree): that read carries the
# s_nop / dependency stall the hardware needs, and MFMAs retire in order.  This test walks the generated code of every
# instantiation and fails if an asm statement touches a register of an MFMA that no ordinary VALU instruction has read yet.
@pytest.fixture(scope="module")
def encode_asm(tmp_path_factory):
    if not os.path.isfile(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa_enc") / "rq_encode.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S",
                           "--cuda-device-only", os.path.join(ROOT, "rayuela.jl_amd", "csrc", "rq_encode.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def _vregs(operand_text):
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", operand_text):
        regs.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", operand_text):
        regs.add(int(a))
    return regs


def _mfma_asm_hazards(lines):
    """(line number, text) of asm-block instructions that touch registers of a not-yet-consumed MFMA"""
    pending = []          # [(dest register set)] of MFMAs no compiler-visible VALU instruction has read yet, oldest first
    in_asm = False
    bad = []
    for no, ln in enumerate(lines):
        s = ln.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        op, _, rest = s.partition(" ")
        rest = rest.split(";")[0]
        if in_asm:
            touched = _vregs(rest)
            if any(touched & d for d in pending):
                bad.append((no, s))
            continue
        if op.startswith("v_mfma"):
            dest = _vregs(rest.split(",")[0])
            pending = [d for d in pending if not (d & dest)] + [dest]      # a chained MFMA on the same accumulator replaces it
            continue
        if op.startswith("v_"):
            srcs = _vregs(",".join(rest.split(",")[1:]))
            for i in range(len(pending) - 1, -1, -1):
                if pending[i] & srcs:
                    pending = pending[i + 1:]          # this MFMA and (in-order retirement) every older one are complete
                    break
    return bad


def test_split_encode_asm_never_reads_unconsumed_mfma_results(encode_asm):
    names = sorted(set(re.findall(r"\n(_ZN2rq22encode_pq_split_kernelILi\d+ELi\d+ELi\d+ELb[01]EEEvNS_9EncParamsE):", encode_asm)))
    assert len(names) >= 24, names        # 8 widths x 4 tile counts x 3 wave counts are instantiated
    checked = 0
    for name in names:
        start = encode_asm.index("\n" + name + ":")
        lines = encode_asm[start:encode_asm.index("s_endpgm", start)].splitlines()
        if not any("v_mfma" in ln for ln in lines):
            continue
        bad = _mfma_asm_hazards(lines)
        assert not bad, "%s: inline asm reads MFMA results before any compiler-visible VALU read:\n%s" % (
            name, "\n".join("%d: %s" % b for b in bad[:6]))
        checked += 1
    assert checked >= 24


@pytest.fixture(scope="module")
def filter_asm(tmp_path_factory):
    if not os.path.isfile(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa_filt") / "rq_encode_filter.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S",
                           "--cuda-device-only", os.path.join(ROOT, "rayuela.jl_amd", "csrc", "rq_encode_filter.hip"), "-o", str(out)],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def test_filter_encode_asm_never_reads_unconsumed_mfma_results(filter_asm):
    """the shipped two-launch encode (rq_encode_filter.hip): the exec-masked copy of the winning tile and the v_cmp / v_addc
    mask builder are the asm statements; neither may touch a register of an MFMA no compiler-visible VALU instruction has read"""
    names = sorted(set(re.findall(r"\n(_ZN2rq23encode_pq_filter_kernelILi\d+ELi\d+ELi\d+ELb[01]EEEvNS_9EncParamsE):", filter_asm)))
    assert len(names) >= 24, names
    checked = 0
    for name in names:
        start = filter_asm.index("\n" + name + ":")
        lines = filter_asm[start:filter_asm.index("s_endpgm", start)].splitlines()
        assert any("v_mfma" in ln for ln in lines)
        bad = _mfma_asm_hazards(lines)
        assert not bad, "%s: inline asm reads MFMA results before any compiler-visible VALU read:\n%s" % (
            name, "\n".join("%d: %s" % b for b in bad[:6]))
        checked += 1
    assert checked >= 24


def test_filter_tile_loop_is_one_basic_block(filter_asm):
    """the point of the round-5 rewrite: between the first and the last MFMA of a sub-quantizer there is no branch and no
    label (SIFT and Deep bench shapes), so the LDS reads of tile t + 1 can be in flight under the MFMAs of tile t"""
    for sub, nw in ((16, 12), (6, 16)):
        name = "_ZN2rq23encode_pq_filter_kernelILi%dELi8ELi%dELb0EEEvNS_9EncParamsE" % (sub, nw)
        start = filter_asm.index("\n" + name + ":")
        lines = [ln.strip() for ln in filter_asm[start:filter_asm.index("s_endpgm", start)].splitlines()]
        mf = [i for i, ln in enumerate(lines) if ln.startswith("v_mfma")]
        assert len(mf) == (24 if sub > 8 else 16), len(mf)
        body = lines[mf[0]:mf[-1]]
        assert not any(ln.startswith("s_cbranch") or ln.startswith(".LBB") for ln in body), \
            [ln for ln in body if ln.startswith("s_cbranch") or ln.startswith(".LBB")][:4]


def test_mfma_hazard_walker_detects_a_planted_hazard():
    """the walker itself: an asm read straight behind the MFMA is flagged, one behind a v_min3 of the same tile is not"""
    unsafe = ["v_mfma_f32_32x32x16_bf16 v[0:15], v[86:89], v[58:61], v[0:15]", ";;#ASMSTART", "v_mov_b64 v[90:91], v[2:3]", ";;#ASMEND"]
    safe = ["v_mfma_f32_32x32x16_bf16 v[0:15], v[86:89], v[58:61], v[0:15]", "v_min3_f32 v40, v0, v1, v2", ";;#ASMSTART",
            "v_mov_b64 v[90:91], v[2:3]", ";;#ASMEND"]
    younger = ["v_mfma_f32_32x32x16_bf16 v[0:15], v[86:89], v[58:61], v[0:15]", "v_mfma_f32_32x32x16_bf16 v[16:31], v[86:89], v[58:61], v[16:31]",
               "v_min3_f32 v40, v0, v1, v2", ";;#ASMSTART", "v_mov_b64 v[90:91], v[18:19]", ";;#ASMEND"]
    assert _mfma_asm_hazards(unsafe) and not _mfma_asm_hazards(safe) and _mfma_asm_hazards(younger)
