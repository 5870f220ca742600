"""ISA lint (CPU, needs only hipcc): properties of the generated gfx950 code that cost measured time when the compiler lost them.

Round 2 found two by reading the ISA: SiLU compiled to a ten-instruction IEEE division (the GroupNorm pass
became VALU-bound), and every ds_read_b128 of the attention kernels sat directly in front of its two MFMAs behind an
s_waitcnt lgkmcnt(0).  These tests cross-compile the three files to assembly and check the hot loops stay as written."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vae_tagger_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")


def _asm(name, tmp_path_factory):
    out = tmp_path_factory.mktemp("isa") / (name + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=fast", "-S", "--cuda-device-only",
           "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-o", str(out), os.path.join(CSRC, name + ".hip")]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return out.read_text().split("\n")


def _functions(lines):
    """{mangled name: [instruction lines]} for every kernel of the file"""
    fns, cur = {}, None
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1); fns[cur] = []
        elif cur is not None:
            t = l.strip()
            if t and not t.startswith((";", ".")):
                fns[cur].append(t)
            if "s_endpgm" in l:
                cur = None
    return fns


@pytest.fixture(scope="module")
def groupnorm_asm(tmp_path_factory):
    return _functions(_asm("groupnorm", tmp_path_factory))


@pytest.fixture(scope="module")
def attention_asm(tmp_path_factory):
    return {**_functions(_asm("attn_qk", tmp_path_factory)), **_functions(_asm("attn_pv", tmp_path_factory))}


def test_groupnorm_apply_has_no_ieee_division(groupnorm_asm):
    apply = {k: v for k, v in groupnorm_asm.items() if "gn_apply_kernel" in k}
    assert len(apply) >= 4
    for name, ins in apply.items():
        ops = [i.split()[0] for i in ins]
        assert not any(o.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")) for o in ops), name
        silu = re.search(r"gn_apply_kernelI\w+?Lb(\d)ELi\dE", name).group(1) == "1"          # <T, SILU, OUT>
        if silu:                                                       # one v_exp_f32 and one v_rcp_f32 per element
            assert ops.count("v_exp_f32_e32") == ops.count("v_rcp_f32_e32") > 0, name


def _mfma_runs_between_full_lgkm_waits(ins):
    """lengths of the MFMA runs that an `s_waitcnt lgkmcnt(0)` (alone or combined) separates, inside the part of the kernel that holds MFMAs"""
    idx = [i for i, l in enumerate(ins) if l.startswith("v_mfma")]
    runs, n = [], 0
    for l in ins[idx[0]:idx[-1] + 1]:
        if l.startswith("v_mfma"):
            n += 1
        elif l.startswith("s_waitcnt") and "lgkmcnt(0)" in l and n:
            runs.append(n); n = 0
    if n:
        runs.append(n)
    return runs


@pytest.mark.parametrize("kernel", ["attn_qk_kernelILi3E", "attn_pv_kernelILi256E", "attn_pv_kernelILi128E"])
def test_attention_lds_reads_are_ahead_of_their_mfmas(attention_asm, kernel):
    (name, ins), = [(k, v) for k, v in attention_asm.items() if kernel in k]
    runs = _mfma_runs_between_full_lgkm_waits(ins)
    # read-ahead of six fragments: a full LDS drain at most every ~12 MFMAs (it was every 2 when the compiler serialised the reads)
    assert sum(runs) / len(runs) >= 8.0, (name, runs[:20])
    # and no drain of the vector-memory counter inside the MFMA stream other than the counted / tile-boundary waits written in the source
    assert not any(re.search(r"scratch_(load|store)", l) for l in ins[[i for i, l in enumerate(ins) if l.startswith("v_mfma")][0]:
                                                                      [i for i, l in enumerate(ins) if l.startswith("v_mfma")][-1]]), name


# ---- round 3: register budgets.  Every MFMA kernel of the path runs two waves per SIMD, i.e. <= 256 VGPRs, and the hot ones sit AT that limit: one
# more live value and the allocator spills dozens of registers into the K-loop (38 spills in the first stride-2 kernel, 70-150 in
# the fp8 attention when an overflow watch was added).  The table is what the committed sources compile to; a spill that creeps in shows up here, on the
# CPU, instead of as a few per cent on the GPU.
_BUDGET = {  # file: {kernel substring: max vgpr_spill_count}
    "conv3x3_halo": {"conv3x3_halo_kernelILi2ELi2ELi0ELi8ELi4ELb0E": 0, "conv3x3_halo_kernelILi2ELi2ELi0ELi8ELi4ELb1E": 0,       # bf16 / fp16 operands
                     "conv3x3_halo_kernelILi2ELi2ELi0ELi16ELi6ELb0E": 33},   # the one-wave-per-SIMD experiment tile: 256 AGPRs + 256 VGPRs, spills outside the loop
    "conv3x3_halo_fp8": {"conv3x3_halo_fp8_kernelILi2ELi1E": 0, "conv3x3_halo_fp8_kernelILi4ELi1E": 0, "conv3x3_halo_fp8_kernelILi2ELi2E": 0},
    "conv3x3_s2_halo": {"conv3x3_s2_halo_kernelILb0E": 3, "conv3x3_s2_halo_kernelILb1E": 3},            # three, in the last chunk's epilogue hand-over, none in the steady-state loop
    "conv3x3_s2_halo_fp8": {"conv3x3_s2_halo_fp8_kernel": 0},
    "attn_fp8": {"attn_qk_fp8_kernelILi1E": 0, "attn_qk_fp8_kernelILi3E": 0, "attn_pv_fp8_kernelILi256E": 0, "attn_pv_fp8_kernelILi128E": 0,
                 "proj_fp8_kernel": 1},
}


@pytest.fixture(scope="module")
def budget_asm(tmp_path_factory):
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(_BUDGET)) as ex:                       # (hipcc runs outside the GIL: ~12 s for the five files together)
        return dict(zip(_BUDGET, ex.map(lambda n: _asm(n, tmp_path_factory), _BUDGET)))


@pytest.mark.parametrize("name", sorted(_BUDGET))
def test_mfma_kernels_keep_two_waves_per_simd_without_spilling(budget_asm, name):
    lines = budget_asm[name]
    # each kernel's metadata block lists its keys alphabetically: .name ... .sgpr_spill_count ... .vgpr_count, .vgpr_spill_count ... ; take them per block
    blocks, cur = [], None
    for l in lines:
        if re.match(r"^\s+- \.\w+:", l):              # a new list item of amdhsa.kernels (or of .args: those carry no .name)
            cur = {}; blocks.append(cur)
        m = re.match(r"^\s+(?:- )?\.(\w+):\s+(\S+)\s*$", l)
        if m and cur is not None:
            cur[m.group(1)] = m.group(2)
    kernels = {b["name"]: b for b in blocks if "name" in b and "vgpr_count" in b}
    for sub, max_spill in _BUDGET[name].items():
        (k, b), = [(k, b) for k, b in kernels.items() if sub in k]
        one_wave = "ELi16ELi6E" in k                                # the one-wave-per-SIMD experiment tile: 256 VGPRs + 256 AGPRs by design
        assert int(b["vgpr_count"]) <= (512 if one_wave else 256), (k, b["vgpr_count"])
        assert int(b["vgpr_spill_count"]) <= max_spill and int(b["sgpr_spill_count"]) == 0, (k, b["vgpr_spill_count"], b["sgpr_spill_count"])


# ---- round 4: packed fp32 with a source op_sel.  In the halo conv kernels `v_pk_add_f32 d, a, p op_sel:[0,1]` (the HIGH register of a pair routed into
# the LOW lane) sometimes read 0.0 instead in lanes 48..63: one element of one GroupNorm partial summed against the wrong pivot, so
# ragged shapes encoded differently run to run (vt_common.h, VT_PIN_PAIR / VT_NO_PACKED_F32).  No kernel of the BUILT
# library may contain such an instruction (op_sel_hi -- low register into the high lane -- is what the compiler emits for pinned pairs, and is stable);
# v_pk_mov_b32, the pair shuffle the compiler builds such operands with, is held to the same rule (gn_apply_f32in_kernel).
def test_no_packed_fp32_instruction_routes_a_source_by_op_sel(tmp_path):
    from vae_tagger_amd import _lib
    objdump = os.path.join(os.path.dirname(os.path.dirname(HIPCC)), "lib", "llvm", "bin", "llvm-objdump")
    if not (os.path.exists(_lib.LIB_PATH) and os.path.exists(objdump)):
        pytest.skip("needs the built library and llvm-objdump")
    so = tmp_path / "lib.so"
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.run([objdump, "--offloading", str(so)], check=True, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE)   # unbundles next to the copy
    objs = sorted(p for p in os.listdir(tmp_path) if p.endswith("gfx950"))
    assert len(objs) >= 10, objs                                   # one code object per translation unit
    packed = bad = 0
    kernel = None
    offenders = set()
    for o in objs:
        for l in subprocess.run([objdump, "-d", str(tmp_path / o)], check=True, stdout=subprocess.PIPE, text=True).stdout.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", l)
            if m:
                kernel = m.group(1)
            elif re.search(r"\bv_pk_(add_f32|mul_f32|fma_f32|mov_b32)\b", l):      # every VOP3P instruction on 64-bit register pairs
                packed += 1
                if re.search(r"op_sel:\[", l):
                    bad += 1; offenders.add(kernel)
    assert packed > 1000                                           # the epilogues do use packed fp32
    assert bad == 0, sorted(offenders)
