import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = os.path.join(ROOT, 'cugraph-gnn_amd/csrc/wg_sage_mfma.hip')
s = open(p).read()
# all waves stamp: stamps[(n * 8 + wave) * 2 + {0: begin, 1: work done}]
s = s.replace("if (a.stamps && blockIdx.x == 0 && wave == CW && lane == 0 && n < 64) a.stamps[(n * 2 + 1) * 3 + 0] = __builtin_readcyclecounter();",
              "if (a.stamps && blockIdx.x == 0 && lane == 0 && n < 64) a.stamps[(n * 8 + wave) * 2 + 0] = __builtin_readcyclecounter();")
s = s.replace("if (a.stamps && blockIdx.x == 0 && wave == CW && lane == 0 && n < 64) a.stamps[(n * 2 + 1) * 3 + 1] = __builtin_readcyclecounter();",
              "if (a.stamps && blockIdx.x == 0 && lane == 0 && n < 64) a.stamps[(n * 8 + wave) * 2 + 1] = __builtin_readcyclecounter();")
s = s.replace("      if (a.stamps && blockIdx.x == 0 && wave == CW && lane == 0 && n < 64) a.stamps[(n * 2 + 1) * 3 + 2] = __builtin_readcyclecounter();\n", "")
s = s.replace("if (a.stamps && blockIdx.x == 0 && wave == 0 && lane == 0 && n < 64) a.stamps[(n * 2) * 3 + 0] = __builtin_readcyclecounter();",
              "if (a.stamps && blockIdx.x == 0 && lane == 0 && n < 64) a.stamps[(n * 8 + wave) * 2 + 0] = __builtin_readcyclecounter();")
s = s.replace("if (a.stamps && blockIdx.x == 0 && wave == 0 && lane == 0 && n < 64) a.stamps[(n * 2) * 3 + 1] = __builtin_readcyclecounter();",
              "if (a.stamps && blockIdx.x == 0 && lane == 0 && n < 64) a.stamps[(n * 8 + wave) * 2 + 1] = __builtin_readcyclecounter();")
s = s.replace("        if (a.stamps && blockIdx.x == 0 && wave == 0 && lane == 0 && n < 64) a.stamps[(n * 2) * 3 + 2] = __builtin_readcyclecounter();\n", "")
open(p, 'w').write(s)
p = os.path.join(ROOT, 'tools/tune/sage_mfma_harness.cpp')
s = open(p).read()
s = s.replace("hipMalloc(&d_stamps, 64 * 6 * 8); hipMemset(d_stamps, 0, 64 * 6 * 8);", "hipMalloc(&d_stamps, 64 * 16 * 8); hipMemset(d_stamps, 0, 64 * 16 * 8);")
i = s.index("    std::vector<unsigned long long> st(64 * 6);")
j = s.index("  }\n  return 0;\n}")
s = s[:i] + '''    std::vector<unsigned long long> st(64 * 16);
    hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost);
    for (int n = 8; n < 11; n++) {
      const unsigned long long t0 = st[(n * 8) * 2];
      printf("step %2d (begin -> work done, ticks since the step began on wave 0):", n);
      for (int w = 0; w < 8; w++) printf("  w%d %lld->%lld", w, (long long)(st[(n * 8 + w) * 2] - t0), (long long)(st[(n * 8 + w) * 2 + 1] - t0));
      printf("   | next step begins %lld\\n", (long long)(st[((n + 1) * 8) * 2] - t0));
    }
''' + s[j:]
open(p, 'w').write(s)
print("ok")
