"""tools/isa_audit.py's parsing on a synthetic listing (the real audit compiles every kernel: minutes, run by hand - DESIGN.md 7)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
isa_audit = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa_audit)

ASM = """
\t.text
_ZN3mma5firstEv:                        ; @_ZN3mma5firstEv
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
.LBB0_1:                                ; =>This Loop Header: Depth=1
\tglobal_load_dword v1, v[2:3], off
.LBB0_2:                                ;   Parent Loop BB0_1 Depth=1
\tflat_load_dword v4, v[5:6]
\ts_waitcnt vmcnt(0) lgkmcnt(0)
\tglobal_store_dwordx4 v[7:8], v[9:12], off nt
\ts_cbranch_execnz .LBB0_2
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
.Lfunc_end0:
_ZN3mma6secondEv:                       ; @_ZN3mma6secondEv
.LBB1_1:
\tds_read_b32 v1, v2
\ts_waitcnt lgkmcnt(0)
\tglobal_store_dword v[3:4], v1, off
\ts_cbranch_execnz .LBB1_1
\ts_endpgm
.Lfunc_end1:
"""


def test_kernels_and_innermost_loops_of_a_listing():
    ks = dict(isa_audit.kernels(ASM))
    assert set(ks) == {"_ZN3mma5firstEv", "_ZN3mma6secondEv"}
    first = ks["_ZN3mma5firstEv"]
    loops = isa_audit.innermost_loops(first)
    assert len(loops) == 1                                    # the outer loop contains the inner one: only the inner is innermost
    a, b = loops[0]
    body = [line.strip() for line in first[a:b + 1]]
    assert any(x.startswith("flat_load") for x in body) and any("vmcnt(0)" in x for x in body) and any(x.startswith("global_store") for x in body)
    second = ks["_ZN3mma6secondEv"]
    (a2, b2), = isa_audit.innermost_loops(second)
    assert not any("vmcnt" in line for line in second[a2:b2 + 1])      # an LDS read ahead of a store waits on lgkmcnt only


def test_the_allow_list_names_the_kernels_that_take_pointer_tables():
    assert isa_audit.ALLOW.search("void mma::adam_kernel<true>(...)") and isa_audit.ALLOW.search("mma::pack_blocks_kernel<false>")
    assert not isa_audit.ALLOW.search("mma::segsum_block_kernel(mma::SegSumParams)")
