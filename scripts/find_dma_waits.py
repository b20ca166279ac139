"""Diagnostic: list compiler-inserted `s_waitcnt vmcnt(..)` (not from inline asm) in kernels that use LDS-DMA.
LLVM's waitcnt pass makes an LDS read that carries a memory operand wait for EVERY outstanding `buffer_load ... lds`
(it cannot prove the read does not alias the DMA's destination), which silently removes a prefetch ring's depth.
usage: find_dma_waits.py file.s"""
import re, sys
name, in_asm, has_dma, waits, n = None, False, False, [], 0
def flush():
    if name and has_dma and waits:
        print(f"{name[:110]}\n   " + "\n   ".join(waits))
for line in open(sys.argv[1]):
    n += 1
    m = re.match(r"^(_Z\w+):", line)
    if m:
        flush(); name, has_dma, waits, in_asm = m.group(1), False, [], False
        continue
    if "#ASMSTART" in line: in_asm = True
    elif "#ASMEND" in line: in_asm = False
    elif "buffer_load" in line and " lds" in line: has_dma = True
    elif "s_waitcnt" in line and "vmcnt" in line and not in_asm:
        waits.append(f"line {n}: {line.strip()}")
flush()
