"""Scan AMDGPU assembly (hipcc -S --cuda-device-only) for a miscompile seen with ROCm 7.2's backend in round 6: a uniform select on a
VECTOR compare emitted as  v_cmp_* vcc ...  ;  s_cselect_b32 ...  with no SCC-defining instruction in between (s_cselect reads SCC, which
v_cmp does not write; the correct sequence has  s_and_b64 sN, vcc|sM, exec  in between).  It turned `Sc[S_PICK] = pick ? 1.0 : 0.0` in
k_scal_step into a store of a stale condition.  Prints every SCC reader (s_cselect, s_cbranch_scc, ...) in a basic block that holds a v_cmp in front of it and NO SCC-defining
instruction -- the shape of the miscompiled select (SCC live-in from a predecessor block would be legal, and rare).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only file.hip -o file.s ;  python tools/scan_scc.py file.s"""
import re, sys
SCC_WRITERS = re.compile(r"^\s*s_(cmp|cmpk|and|or|xor|andn2|orn2|nand|nor|xnor|add|sub|addc|subb|min|max|lshl|lshr|ashr|bfe|bfm|absdiff|abs|not|bcnt|ff|flbit|bitcmp|mul_hi|lshl\d_add|quadmask|wqm|sext|brev|and_saveexec|or_saveexec|andn\d_saveexec|mov_b32 scc|cmov)")
NOT_SCC = re.compile(r"^\s*s_(mov|movk|cselect|cbranch|branch|waitcnt|nop|load|buffer_load|barrier|sleep|getreg|setreg|memrealtime|memtime|endpgm|mul_i32|sendmsg|setprio|dcache|icache|getpc|swappc|setpc|call|trap|bitset|pack|cvt|movrel|subvector|ttrace|version|code_end|inst_prefetch|clause|round_mode|denorm_mode|waitcnt_|delay)")
bad = 0
for path in sys.argv[1:]:
    fn, scc_written, vcmp_line = None, False, 0          # per basic block: an SCC writer seen? the last v_cmp's line
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        if t.startswith("; %bb."):                            # a fall-through block
            scc_written, vcmp_line = False, 0
            continue
        if not t or t.startswith(";"):
            continue
        if t.endswith(":"):                                   # a label: new basic block (SCC may be live-in from a predecessor: unknown)
            scc_written, vcmp_line = False, 0
            if t.startswith("_Z"):
                fn = t[:-1]
            continue
        if t.startswith("."):
            continue
        op = t.split()[0]
        if op.startswith("v_cmp"):
            vcmp_line = n
        elif op in ("s_cselect_b32", "s_cselect_b64", "s_cbranch_scc0", "s_cbranch_scc1", "s_cmov_b32", "s_addc_u32", "s_subb_u32"):
            # reads SCC.  Suspicious: nothing in this block has written SCC, yet a v_cmp (which writes VCC / an SGPR pair, not SCC)
            # stands in front of it -- the shape of the miscompiled select
            if not scc_written and vcmp_line:
                print("%s:%d: %s reads SCC; no SCC writer in its basic block, a v_cmp at line %d  [%s]" % (path, n, op, vcmp_line, fn))
                bad += 1
        elif op.startswith("s_") and SCC_WRITERS.match(t):
            scc_written = True
print("%d suspicious sites" % bad)
sys.exit(1 if bad else 0)
