"""Print the instruction stream of the MFMA loops of one kernel as a compact string, one line per basic block
(build container, no GPU needed):
    python tools/isa_loop.py vilbert-multi-task_amd/csrc/gemm_planes.hip gemm_planes_kernelILb1ELb1ELb1ELi3 -DVB_NPL=3
Legend: M mfma, r ds_read, w ds_write, G / g global_load_dwordx4 / dword, c cvt_pk_bf16, s v_sub, l v_lshl,
a v_and, n s_nop, B s_barrier, W(..) s_waitcnt, ? branch, . anything else. Also prints VGPR / scratch usage.
This is how the issue order of the K loops (prefetch placement, MFMA / VALU interleave, exposed waits) was
checked while tuning."""
import re
import subprocess
import sys
import tempfile

T = {"ds_read_b128": "r", "ds_read_b32": "r", "ds_read_b64": "r", "ds_write_b128": "w", "ds_write_b64": "w",
     "ds_write_b32": "w", "ds_write2st64_b64": "w", "s_barrier": "B", "global_load_dwordx4": "G",
     "global_load_dword": "g", "s_nop": "n", "v_cvt_pk_bf16_f32": "c", "v_sub_f32_e32": "s",
     "v_lshlrev_b32_e32": "l", "v_and_b32_e32": "a"}


def main():
    src, pattern, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    with tempfile.NamedTemporaryFile(suffix=".s") as f:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics",
                        "-S", "--cuda-device-only", src, "-o", f.name] + extra, check=True,
                       stderr=subprocess.DEVNULL)
        s = open(f.name).read()
    names = sorted(set(re.findall(r"^(_Z\w*%s\w*):" % re.escape(pattern), s, flags=re.M)))
    for name in names:
        i = s.index(name + ":")
        body = s[i:s.index(".Lfunc_end", i)].split("\n")
        print("==", name)
        cur, label = [], None
        blocks = []
        for line in body:
            if line.startswith(".LBB"):
                blocks.append((label, cur))
                label, cur = line.split(":")[0], []
                continue
            m = re.match(r"\s+([a-z_0-9]+)\s*(.*)", line)
            if not m:
                continue
            op = m.group(1)
            if op.startswith("v_mfma"):
                t = "M"
            elif op == "s_waitcnt":
                t = "W(" + m.group(2).strip().replace("cnt", "") + ")"
            elif op.startswith("s_cbranch") or op == "s_branch":
                t = "?"
            else:
                t = T.get(op, ".")
            cur.append(t)
        blocks.append((label, cur))
        for label, cur in blocks:
            if sum(1 for t in cur if t == "M") >= 8:
                print("%-12s %4d  %s" % (label, len(cur), "".join(cur)))
        meta = s[s.index(".name:           " + name):][:4000]
        for key in (".vgpr_count", ".agpr_count", ".private_segment_fixed_size", ".group_segment_fixed_size"):
            m = re.search(key + r":\s+(\d+)", meta)
            if m:
                print("   %s = %s" % (key, m.group(1)))


if __name__ == "__main__":
    main()
