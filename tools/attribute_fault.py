"""Attribute the frames of a crash log to modules with the fault-time /proc/self/maps (MBFIR_FAULT_MAPS, csrc/api.cpp).
    python tools/attribute_fault.py <fault_maps.txt> <crash log with '@ 0x...' frames>
Prints module + file offset per frame, the mapping the fault address belongs to (or the mapping that ENDS at it), and --
when the module exists on this machine and llvm-objdump is there -- the faulting instruction."""
import os, re, subprocess, sys

maps = []
fault = None
for ln in open(sys.argv[1]):
    m = re.match(r"([0-9a-f]+)-([0-9a-f]+) (\S+) ([0-9a-f]+) \S+ \d+\s*(.*)", ln)
    if m:
        maps.append((int(m.group(1), 16), int(m.group(2), 16), m.group(3), int(m.group(4), 16), m.group(5).strip()))
    m = re.search(r"mbfir fault: signal (0x[0-9a-f]+) at address (0x[0-9a-f]+)", ln)
    if m:
        fault = int(m.group(2), 16)


def where(a):
    for lo, hi, pr, off, name in maps:
        if lo <= a < hi:
            return name or "[anonymous]", pr, a - lo + off, lo, hi
    return None


frames = [int(x, 16) for x in re.findall(r"@\s+(0x[0-9a-f]+)", open(sys.argv[2]).read())]
print("fault address %#x" % fault)
w = where(fault)
if w:
    print("  inside %s %s (%#x-%#x)" % (w[0], w[1], w[3], w[4]))
else:
    for lo, hi, pr, off, name in maps:
        if hi == fault:
            print("  UNMAPPED; it is the first byte past the mapping %#x-%#x %s %s (%d KiB)" % (lo, hi, pr, name, (hi - lo) // 1024))
first = True
for pc in frames:
    w = where(pc)
    if not w:
        print("%#x  unmapped" % pc)
        continue
    print("%#x  %s  file offset %#x" % (pc, os.path.basename(w[0]), w[2]))
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if first and os.path.exists(w[0]) and os.path.exists(objdump) and "librocprofiler" in w[0]:
        first = False
        out = subprocess.run([objdump, "-d", "--start-address=%#x" % (w[2] - 7), "--stop-address=%#x" % (w[2] + 9), w[0]], capture_output=True, text=True).stdout
        for ln in out.splitlines():
            if re.match(r"\s+[0-9a-f]+:", ln):
                print("        " + ln.strip())
