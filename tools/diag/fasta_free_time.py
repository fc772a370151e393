"""Where the 0.2 s after the FASTA ingest goes: the time of phk_fasta_free() of a 5 GB sequence buffer, alone and after
the upload (which pins the buffer), and whether the buffer is made of transparent huge pages.
    python tools/diag/fasta_free_time.py [contigs]"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from phamers_amd import _lib
import e2e_fasta


def rollup():
    out = {}
    for line in open("/proc/self/smaps_rollup"):
        p = line.split()
        if p[0] in ("Rss:", "AnonHugePages:", "Anonymous:"):
            out[p[0][:-1]] = int(p[1]) >> 10
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    ctx = _lib.get_context()
    root = tempfile.mkdtemp(prefix="phk_free_")
    path = os.path.join(root, "c.fasta")
    e2e_fasta.write_fasta(ctx, path, n, 5000)
    try:
        for upload, nt in ((False, 8), (True, 8), (True, 0), (True, 8)):
            os.environ["PHK_FREE_THREADS"] = str(nt)
            t = time.perf_counter(); f = _lib.Fasta(path); t_read = time.perf_counter() - t
            mem = rollup()
            t_up = None
            if upload:
                t = time.perf_counter(); b = _lib.Batch.from_fasta(ctx, f, 4); ctx.sync(); t_up = time.perf_counter() - t
            mem_up = rollup()
            t = time.perf_counter(); f.close(); t_free = time.perf_counter() - t
            if upload:
                b.close()
            print({"read_s": round(t_read, 3), "upload_count_s": t_up and round(t_up, 3), "free_s": round(t_free, 3), "threads": nt, "MiB": mem, "THP_after_upload": mem_up["AnonHugePages"]})
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
