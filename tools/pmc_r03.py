"""Table of the three PMC passes of tools/pmc_traffic.sh (fetch / write / util) over the GEMM lab: consecutive dispatches
of the same kernel + grid are merged (the lab times every shape with 23 launches of forward, dgrad, wgrad in turn).
    python tools/pmc_r03.py gpurun_out/<name>      -> stdout"""
import ast
import re
import sys


def load(path):
    out = []
    for l in open(path):
        m = re.match(r"dispatch (\d+): (.*)", l)
        if m:
            e = ast.literal_eval(m.group(2))
            if e.get("dur_us"):
                out.append(e)
    return out


def groups(rows):
    g, cur = [], None
    for e in rows:
        key = (e["kernel"], e["grid"])
        if cur is None or cur[0] != key:
            cur = [key, []]
            g.append(cur)
        cur[1].append(e)
    return g


def main():
    base = sys.argv[1]
    fetch, write, util = (groups(load("%s_%s.txt" % (base, t))) for t in ("fetch", "write", "util"))
    print("kernel / blocks / launches | FETCH_SIZE KB (x2 = bytes read, gfx950) | WRITE_SIZE KB | us | clk GHz | mfma_busy")
    for gf, gw, gu in zip(fetch, write, util):
        n = len(gu[1])
        f = sum(e.get("FETCH_SIZE", 0) for e in gf[1]) / max(len(gf[1]), 1)
        w = sum(e.get("WRITE_SIZE", 0) for e in gw[1]) / max(len(gw[1]), 1)
        dur = sum(e["dur_us"] for e in gu[1]) / n
        cyc = sum(e.get("GRBM_GUI_ACTIVE", 0) for e in gu[1]) / n / 8
        busy = sum(e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for e in gu[1]) / n / (1024 * cyc) if cyc else 0
        print("%-48s blocks %5d x %2d  FETCH %9.0f KB  WRITE %9.0f KB  %8.1f us  clk %.2f GHz  mfma_busy %.3f" % (
            gu[0][0], gu[0][1], n, f, w, dur, cyc / dur / 1e3, busy))


if __name__ == "__main__":
    main()
