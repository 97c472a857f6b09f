"""kNN / FPS timings at the BASELINE shapes (the index_ops block of bench.py on its own).  python tools/knn_bench.py"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for B, N, S, k in ((256, 1024, 256, 16), (256, 2048, 512, 16), (8, 8192, 2048, 16), (128, 2048, 196, 32)):
    r = bench.index_ops_roofline(B, N, S, k, torch.device('cuda'))
    print(f"B={B} N={N} S={S} k={k}: FPS {r['fps']['us']:.0f} us, kNN {r['knn']['us']:.0f} us ({r['knn']['pair_distance_TFLOPs']:.2f} TFLOP/s of pair distances)")
