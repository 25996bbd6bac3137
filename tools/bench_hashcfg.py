#!/usr/bin/env python3
"""Complete proofs (mh_prove) of miden:LOG_N:51:8 under one of the reference's StarkConfigs; for rocprofv3 runs.
    python tools/bench_hashcfg.py [blake3|keccak|rpo|rpx|poseidon2] [--log-n 20] [--steps 4]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lmcs", nargs="?", default="blake3")
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--steps", type=int, default=4)
    a = ap.parse_args()
    import bench
    from __graft_entry__ import load_package
    pkg = load_package()
    print(json.dumps(dict(lmcs=a.lmcs, log_n=a.log_n, **bench.hash_config_probe(pkg, 0, a.log_n, a.lmcs, steps=a.steps))))


if __name__ == "__main__":
    main()
