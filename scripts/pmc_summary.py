#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of scripts/gpu_profile.sh (gpurun_out/pmc/p1..p5) into the per-kernel table committed
under profiles/ (*_pmc_summary.txt), which bench.py reads for roofline.traffic.

usage: pmc_summary.py <pmc_dir> <out.txt> [title]
Columns: calls, average duration (kernel trace of pass p1), shader clock implied by GRBM_GUI_ACTIVE (summed over the 8 XCDs),
MFMA busy % (SQ_VALU_MFMA_BUSY_CYCLES, a sum over the 1024 SIMDs, over kernel cycles x 1024), wave cycles / parked / issue-stalled cycles, LDS bank-conflict cycles, FETCH_SIZE and WRITE_SIZE in KB per dispatch
(FETCH_SIZE counts wide coalesced reads at half size on gfx950: HBM bytes = 2 x FETCH + WRITE, MI355X_MICROARCH.md)."""
import csv
import glob
import sys
from collections import defaultdict


def load(pmc_dir, p):
    rows = defaultdict(lambda: defaultdict(list))
    for path in glob.glob('%s/%s/*counter_collection.csv' % (pmc_dir, p)):
        for r in csv.DictReader(open(path)):
            rows[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return rows


def durations(pmc_dir, p):
    d = defaultdict(list)
    for path in glob.glob('%s/%s/*kernel_trace.csv' % (pmc_dir, p)):
        for r in csv.DictReader(open(path)):
            d[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    return d


def short(name):
    """kernel name as one token: no 'void ', no argument list, no spaces (bench.py looks it up by this token)"""
    name = name.replace('(RyIgemmParams)', '').replace('(RyReduceParams)', '')
    i = name.find('(')
    name = name if i < 0 else name[:i]
    if name.startswith('void '):
        name = name[5:]
    return name.replace(' ', '')[:63]


def main(pmc_dir, out, title):
    P = {p: load(pmc_dir, p) for p in ('p1', 'p2', 'p3', 'p4', 'p5')}
    dur = durations(pmc_dir, 'p1')
    mean = lambda v: sum(v) / len(v) if v else 0.0
    total = lambda v: sum(v) if v else 0.0
    names = sorted(dur, key=lambda n: -sum(dur[n]))
    with open(out, 'w') as f:
        f.write('# %s\n' % title)
        f.write('# GRBM_GUI_ACTIVE is summed over the 8 XCDs (clk = GUI / 8 / duration); FETCH_SIZE/WRITE_SIZE in KB per dispatch, '
                'FETCH_SIZE reads 1/2 of wide coalesced streams on gfx950 (MI355X_MICROARCH.md)\n')
        f.write('%-64s %8s %10s %8s %12s %12s %12s %12s %10s %12s %12s\n' % (
            'kernel', 'calls', 'avg_us', 'clk_GHz', 'mfma_busy%', 'wave_cyc', 'wait_any', 'wait_inst', 'bank_conf', 'fetch_KB', 'write_KB'))
        for n in names:
            c1, c2, c3, c4, c5 = (P[p].get(n, {}) for p in ('p1', 'p2', 'p3', 'p4', 'p5'))
            avg = mean(dur[n])
            gui = mean(c1.get('GRBM_GUI_ACTIVE', []))
            clk = gui / 8.0 / (avg * 1e3) if avg else 0.0
            # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (64 cycles per v_mfma_f32_32x32x2_f32); GUI / 8 = kernel cycles
            mfma = 100.0 * mean(c1.get('SQ_VALU_MFMA_BUSY_CYCLES', [])) / (gui / 8.0 * 1024.0) if gui else 0.0
            f.write('%-64s %8d %10.1f %8.2f %12.1f %12.3g %12.3g %12.3g %10.3g %12.0f %12.0f\n' % (
                short(n), len(dur[n]), avg, clk, mfma, total(c1.get('SQ_WAVE_CYCLES', [])), total(c2.get('SQ_WAIT_ANY', [])),
                total(c2.get('SQ_WAIT_INST_ANY', [])), total(c3.get('SQ_LDS_BANK_CONFLICT', [])),
                mean(c4.get('FETCH_SIZE', [])), mean(c5.get('WRITE_SIZE', []))))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else 'rocprofv3 --pmc summary')
