"""Per-SHAPE table + JSON of the round-6 counter passes over tools/bf16_lab time (tools/pmc_r06.sh): the lab launches every shape
23 times (3 warm-up + 20 timed); the last 20 dispatches of each group are averaged. Weight-gradient shapes are PAIRS of launches
(wgrad_bf16_kernel + wgrad_bf16_reduce_kernel, the default deterministic form): both are listed and summed.
    python tools/pmc_r06_table.py gpurun_out/r06_pmc_bf16 profiles/r06_bf16_gemm_traffic.json > profiles/r06_gemm_pmc.txt"""
import ast
import json
import re
import sys

LIN = [("text q|k|v fwd", 9216, 2304, 768, 0), ("text attn-out fwd (+drop+res)", 9216, 768, 768, 2),
       ("text FFN up fwd (GELU)", 9216, 3072, 768, 1), ("text FFN down fwd (+drop+res)", 9216, 768, 3072, 2),
       ("text FFN down dgrad (x gelu')", 9216, 3072, 768, 3), ("text FFN up dgrad (+res)", 9216, 768, 3072, 4),
       ("text q|k|v dgrad", 9216, 768, 2304, 0), ("image q|k|v fwd", 9472, 3072, 1024, 0),
       ("image 1024 fwd (+drop+res)", 9472, 1024, 1024, 2), ("image FFN up fwd (GELU)", 9472, 1024, 1024, 1),
       ("image q|k|v dgrad", 9472, 1024, 3072, 0), ("co-attn text q|k|v fwd", 9216, 3072, 768, 0),
       ("co-attn out -> text (+drop+res)", 9216, 768, 1024, 2), ("image features fwd", 9472, 1024, 2048, 0),
       ("B=64 text q|k|v fwd", 2304, 2304, 768, 0), ("B=64 image 1024 fwd", 2368, 1024, 1024, 2)]
WG = [("text q|k|v", 9216, 2304, 768), ("text attn-out", 9216, 768, 768), ("text FFN up", 9216, 3072, 768),
      ("text FFN down", 9216, 768, 3072), ("image q|k|v", 9472, 3072, 1024), ("image 1024", 9472, 1024, 1024),
      ("co-attn text q|k|v", 9216, 3072, 768), ("co-attn out -> text", 9216, 768, 1024), ("image features", 9472, 1024, 2048),
      ("B=64 text FFN up", 2304, 3072, 768), ("B=64 image 1024", 2368, 1024, 1024)]


def load(path):
    out = []
    for l in open(path):
        m = re.match(r"dispatch (\d+): (.*)", l)
        if m:
            out.append(ast.literal_eval(m.group(2)))
    return out


def avg(rows, key):
    return sum(r.get(key, 0.0) for r in rows) / max(len(rows), 1)


def main():
    base, jpath = sys.argv[1], sys.argv[2]
    fetch, write, util = (load("%s_%s.txt" % (base, t)) for t in ("fetch", "write", "util"))
    lin = [i for i, e in enumerate(util) if "gemm_bf16_kernel" in e["kernel"]]
    wgk = [i for i, e in enumerate(util) if "wgrad_bf16_kernel" in e["kernel"]]
    red = [i for i, e in enumerate(util) if "reduce" in e["kernel"]]
    assert len(lin) == 23 * len(LIN) and len(wgk) == 23 * len(WG), (len(lin), len(wgk), len(red))
    print("Round 6 counter evidence for the bf16 training kernels (tools/pmc_r06.sh: three separate rocprofv3 --pmc passes over "
          "tools/bf16_lab time;\nper shape = mean of the 20 timed launches; FETCH_SIZE x 2 = bytes read through the fabric on "
          "gfx950 (MI355X_MICROARCH.md, HBM section),\nInfinity-Cache hits included; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
          "(1024 SIMDs x GRBM_GUI_ACTIVE / 8); algorithmic = operands once + outputs once)\n")
    launches = []

    def row(name, M, N, K, idx, alg, kern):
        f = avg([fetch[i] for i in idx], "FETCH_SIZE") * 2 * 1024
        w = avg([write[i] for i in idx], "WRITE_SIZE") * 1024
        u = [util[i] for i in idx]
        dur = avg(u, "dur_us")
        cyc = avg(u, "GRBM_GUI_ACTIVE") / 8
        busy = avg(u, "SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * cyc) if cyc else 0.0
        print("%-32s %5d x %4d x %4d  %-22s %7.1f us  mfma_busy %.3f  read %7.1f MB  write %7.1f MB  = %7.1f MB vs %6.1f MB algorithmic (%.2fx)"
              % (name, M, N, K, kern, dur, busy, f / 1e6, w / 1e6, (f + w) / 1e6, alg / 1e6, (f + w) / alg))
        return {"shape": name, "kernel": kern, "M": M, "N": N, "K": K, "us": round(dur, 1), "mfma_busy": round(busy, 3),
                "read_bytes_corrected": f, "write_bytes": w, "hbm_bytes_corrected": f + w, "algorithmic_bytes": alg}
    print("== vb_linear_bf16: gemm_bf16_kernel (forward / dgrad through the transposed shadow)")
    for s, (name, M, N, K, epi) in enumerate(LIN):
        idx = lin[23 * s + 3:23 * s + 23]
        outs = 2 if epi == 1 else 1
        extra = M * N * 2 if epi in (2, 3, 4) else 0
        alg = (M * K + N * K) * 2 + outs * M * N * 2 + extra
        launches.append(row(name, M, N, K, idx, alg, "gemm_bf16_kernel"))
    print("\n== vb_wgrad_bf16 (deterministic default): wgrad_bf16_kernel (partials) + wgrad_bf16_reduce_kernel (ordered reduce into dW)")
    have_red = len(red) > 0
    r_pos = 0
    for s, (name, M, N, K) in enumerate(WG):
        idx = wgk[23 * s + 3:23 * s + 23]
        alg = (M * N + M * K) * 2 + N * K * 4 * 2          # operands once, dW read + written once (the kernel ADDS)
        a = row(name, M, N, K, idx, alg, "wgrad_bf16_kernel")
        # the reduce launches that directly follow those wgrad dispatches
        ridx = [i + 1 for i in idx if i + 1 < len(util) and "reduce" in util[i + 1]["kernel"]]
        if ridx:
            b = row("  + ordered reduce", M, N, K, ridx, alg, "wgrad_bf16_reduce_kernel")
            a["reduce"] = b
            a["pair_hbm_bytes_corrected"] = a["hbm_bytes_corrected"] + b["hbm_bytes_corrected"]
            a["pair_us"] = round(a["us"] + b["us"], 1)
            print("  = pair %7.1f us, %7.1f MB vs %6.1f MB algorithmic (%.2fx)" % (a["pair_us"], a["pair_hbm_bytes_corrected"] / 1e6,
                                                                                 alg / 1e6, a["pair_hbm_bytes_corrected"] / alg))
        launches.append(a)
    json.dump({"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / utilisation (three separate passes, "
                      "tools/pmc_r06.sh -> tools/pmc_traffic.sh) over tools/bf16_lab time (product library, round-6 binaries); "
                      "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 read correction, MI355X_MICROARCH.md)",
               "launches": launches}, open(jpath, "w"), indent=1)


if __name__ == "__main__":
    main()
