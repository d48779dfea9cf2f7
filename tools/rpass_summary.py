"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr saved to a file):
one line per kernel with VGPRs, SGPRs, scratch, occupancy and spills."""
import re
import subprocess
import sys


def main(path, pat=None):
    txt = open(path).read()
    blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
    for b in blocks:
        name = b.split('\n')[0].strip()

        def g(k):
            m = re.search(k + r': (\d+)', b)
            return int(m.group(1)) if m else -1
        try:
            n = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        except FileNotFoundError:
            n = name
        n = re.sub(r'^void ', '', n)
        n = re.sub(r'\(.*', '', n)
        if pat and not re.search(pat, n):
            continue
        print("%-64s V=%3d S=%3d scr=%4d occ=%d sSpill=%3d vSpill=%3d lds=%d" % (
            n[:64], g('VGPRs'), g('SGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'),
            g('SGPRs Spill'), g('VGPRs Spill'), g(r'LDS Size \[bytes/block\]')))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
