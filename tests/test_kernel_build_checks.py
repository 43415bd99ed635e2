"""Build-time checks of the hand-ordered GEMM K loop: csrc/gemm4.hip writes its K loop as `asm volatile` statements that the
compiler's scheduler, waitcnt insertion and hazard recogniser cannot see into -- correctness and speed rest on hipcc only ALLOCATING registers
there.  This test cross-compiles the file for gfx950 (no GPU needed), finds the steady-state K loop of every gemm4 kernel in the assembly (the
basic block with 256 MFMAs that branches back to itself: two K-tiles) and asserts that it is exactly the stream the source spells out: nothing
but the 256 MFMAs, the 64 fragment reads, the 32 LDS-DMA pieces with their M0 set-up, the counted waits and four barriers -- no scratch
access, no v_accvgpr / v_mov traffic (a spill or a copy of a fragment register between its ds_read and its s_waitcnt would read stale data
silently), no vector ALU at all."""
import collections
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"

ALLOWED = {"v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x32_bf16", "ds_read_b128", "global_load_lds_dwordx4", "s_add_u32", "s_addc_u32", "s_add_i32",
           "s_nop", "s_waitcnt", "s_barrier", "s_cmp_ge_i32", "s_cmp_lt_i32", "s_cbranch_scc0", "s_cbranch_scc1", "s_mov_b32", "s_mov_b64"}


@pytest.fixture(scope="module")
def gemm4_asm(tmp_path_factory):
    if not shutil.which(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("asm") / "gemm4.s"
    src = os.path.join(ROOT, "dinov2.cpp_amd", "csrc", "gemm4.hip")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-lambda-capture", "-x", "hip", "-S", "--cuda-device-only", src, "-o", str(out)],
                   check=True, capture_output=True, timeout=600)
    return out.read_text()


def _functions(txt):
    for f in re.split(r"\n(?=_ZN6dinov2\w+:\s)", txt):
        name = f.split(":", 1)[0]
        if name.startswith("_ZN6dinov2") and "gemm4" in name and "kernel" in name:
            yield name, f


def _blocks(body):
    lab, cur = None, []
    for line in body.splitlines():
        s = line.strip()
        if re.match(r"^\.LBB\d+_\d+:", s):
            if cur:
                yield lab, cur
            lab, cur = s.split(":")[0], []
        elif s and not s.startswith(";") and not s.startswith("."):
            cur.append(s.split(";")[0].strip())
    if cur:
        yield lab, cur


def test_gemm4_steady_state_k_loop_is_the_hand_written_stream(gemm4_asm):
    seen = 0
    for name, body in _functions(gemm4_asm):
        loops = [(lab, b) for lab, b in _blocks(body)
                 if sum(x.startswith("v_mfma") for x in b) >= 200 and any(x.startswith("s_cbranch") and lab in x for x in b)]
        if "Li8EEE" not in name and "mixed" not in name:
            continue  # (short one-tile-per-workgroup instantiations, NI = 2 / 3 / 4: their K-tiles have 32 - 64 MFMAs; same macros)
        assert len(loops) >= 1, name
        for lab, b in loops:
            ops = collections.Counter(x.split()[0] for x in b)
            assert set(ops) <= ALLOWED, (name, lab, sorted(set(ops) - ALLOWED))
            nm = ops["v_mfma_f32_16x16x32_f16"] + ops["v_mfma_f32_16x16x32_bf16"]
            assert nm in (256, 192), (name, lab, nm)  # two K-tiles of the 256-row body (128 MFMAs each) or of the 192-row body (96)
            assert ops["ds_read_b128"] == (64 if nm == 256 else 56), (name, lab, ops["ds_read_b128"])
            assert ops["global_load_lds_dwordx4"] == (32 if nm == 256 else 28), (name, lab, ops["global_load_lds_dwordx4"])
            assert ops["s_barrier"] == 4 and ops["s_waitcnt"] == 6, (name, lab, dict(ops))
            seen += 1
    assert seen >= 20  # 2 dtypes x 5 epilogues x (plain kernel: one loop; mixed kernel: the 256- and the 192-row body)


def test_gemm4_kernels_use_the_whole_register_file_and_little_scratch(gemm4_asm):
    """512 registers per lane (256 accumulators + fragments): one wave per SIMD by construction.  Scratch: the residual epilogue keeps three
    passes of residual rows in flight next to the next tile's fragments and spills a handful of registers OUTSIDE the K loop (checked
    above); anything beyond that would be a regression of the register budget."""
    n = 0
    for m in re.finditer(r"\.amdhsa_kernel (_ZN6dinov2\w+)(.*?)\.end_amdhsa_kernel", gemm4_asm, re.S):
        name, desc = m.group(1), m.group(2)
        if "gemm4" not in name:
            continue
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", desc).group(1))
        nvgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", desc).group(1))
        assert scratch <= 64, (name, scratch)
        if "Li8EEE" in name or "mixed" in name:
            assert 440 <= nvgpr <= 512, (name, nvgpr)
        n += 1
    assert n >= 20


def _regs(tok):
    """VGPR numbers named by an operand token: v12 -> {12}, v[24:27] -> {24, 25, 26, 27}; anything else -> {}."""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def test_gemm4_no_instruction_touches_a_fragment_before_its_wait(gemm4_asm):
    """The fragment reads are `asm volatile("ds_read_b128 ...")` statements whose s_waitcnt is ANOTHER statement: to the compiler the output
    register holds its value the moment the statement ends.  Anything it places in between that reads or moves such a register -- a spill,
    an AGPR park, a copy -- picks up whatever the register held BEFORE the LDS data arrived.  Round 6 hit exactly that: under the register
    pressure of the first EPI_QKV_LN epilogue hipcc spilled Pw[3] right behind its ds_read (scratch_store two lines later, the wait after it);
    every tile lost its first k-step in sixteen columns.  Checked here for every gemm4 kernel, over the whole function: between a
    ds_read_b128 and the next `s_waitcnt lgkmcnt(0)` no other instruction may name the destination registers."""
    checked = 0
    for name, body in _functions(gemm4_asm):
        pending = set()
        for line in body.splitlines():
            s = line.split(";")[0].strip()
            if not s or s.startswith(".") or s.endswith(":"):
                continue
            op, _, rest = s.partition(" ")
            toks = [t.strip() for t in rest.split(",")] if rest else []
            if op == "s_waitcnt" and "lgkmcnt(0)" in rest:
                pending.clear()
                continue
            if op == "ds_read_b128":
                pending |= _regs(toks[0])
                checked += 1
                continue
            if pending:
                used = set()
                for t in toks:
                    used |= _regs(t.split(" ")[0])
                assert not (used & pending), (name, s, sorted(used & pending))
    assert checked > 1000
