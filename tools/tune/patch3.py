import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = os.path.join(ROOT, 'cugraph-gnn_amd/csrc/wg_sage_mfma.hip')
s = open(p).read()
s = s.replace("                             // 8 weight fragments of k-step 0 only, 16 / 32 s_setprio 3 for producers / consumers\n};",
              "                             // 8 weight fragments of k-step 0 only, 16 / 32 s_setprio 3 for producers / consumers\n"
              "  unsigned long long* stamps;  // tuning harness only: s_memtime stamps of workgroup 0 [step][role][begin, work done, barrier passed]\n};")
# producer stamps
s = s.replace('''    for (int64_t n = 0; n <= mine; n++) {
      if (n < mine && !(a.debug & 2)) {
        uint32_t* tile_lds = lds + (n & 1) * tile_dw;
        p.load_bounds(tile_of(n + 1), b_next);''', '''    for (int64_t n = 0; n <= mine; n++) {
      if (a.stamps && blockIdx.x == 0 && threadIdx.x == CW * 64 && n < 64) a.stamps[(n * 2 + 1) * 3 + 0] = __builtin_readcyclecounter();
      if (n < mine && !(a.debug & 2)) {
        uint32_t* tile_lds = lds + (n & 1) * tile_dw;
        p.load_bounds(tile_of(n + 1), b_next);''')
s = s.replace('''          for (int it = 0; it < kDepth - 1; it++) p.issue(cur, it, buf[it]);
        }
      }
      lds_barrier();
    }
  } else {''', '''          for (int it = 0; it < kDepth - 1; it++) p.issue(cur, it, buf[it]);
        }
      }
      if (a.stamps && blockIdx.x == 0 && threadIdx.x == CW * 64 && n < 64) a.stamps[(n * 2 + 1) * 3 + 1] = __builtin_readcyclecounter();
      lds_barrier();
      if (a.stamps && blockIdx.x == 0 && threadIdx.x == CW * 64 && n < 64) a.stamps[(n * 2 + 1) * 3 + 2] = __builtin_readcyclecounter();
    }
  } else {''')
s = s.replace('''      for (int64_t n = 0; n <= mine; n++) {
        if (n >= 1 && !(a.debug & 1)) cons.tile(a, tile_of(n - 1), lds + ((n - 1) & 1) * tile_dw, wave, lane, scratch);
        lds_barrier();
      }''', '''      for (int64_t n = 0; n <= mine; n++) {
        if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0 && n < 64) a.stamps[(n * 2) * 3 + 0] = __builtin_readcyclecounter();
        if (n >= 1 && !(a.debug & 1)) cons.tile(a, tile_of(n - 1), lds + ((n - 1) & 1) * tile_dw, wave, lane, scratch);
        if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0 && n < 64) a.stamps[(n * 2) * 3 + 1] = __builtin_readcyclecounter();
        lds_barrier();
        if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0 && n < 64) a.stamps[(n * 2) * 3 + 2] = __builtin_readcyclecounter();
      }''')
s = s.replace("(2 * F + 15) / 16, bias, relu, out, ldo, row_stride_dw(F), 0};", "(2 * F + 15) / 16, bias, relu, out, ldo, row_stride_dw(F), 0, nullptr};")
open(p, 'w').write(s)

p = os.path.join(ROOT, 'tools/tune/sage_mfma_harness.cpp')
s = open(p).read()
s = s.replace("d_bias, 1, d_out, N, row_stride_dw(F), 0};", "d_bias, 1, d_out, N, row_stride_dw(F), 0, nullptr};\n  unsigned long long* d_stamps; hipMalloc(&d_stamps, 64 * 6 * 8); hipMemset(d_stamps, 0, 64 * 6 * 8);")
s = s.replace('''  return 0;
}''', '''  if (argc > 3) {   // timeline of workgroup 0 (shader clock ticks)
    a.debug = atoi(argv[2]); a.stamps = d_stamps;
    launch<void, 32, 64, 4, WG_HARNESS_KSC>(a, true, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st(64 * 6);
    hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost);
    const unsigned long long t0 = st[4 * 6];
    for (int n = 4; n < 12; n++)
      printf("step %2d  consumer: begin %7lld work-done %7lld barrier %7lld | producer: begin %7lld work-done %7lld barrier %7lld\\n", n,
             (long long)(st[n * 6 + 0] - t0), (long long)(st[n * 6 + 1] - t0), (long long)(st[n * 6 + 2] - t0),
             (long long)(st[n * 6 + 3] - t0), (long long)(st[n * 6 + 4] - t0), (long long)(st[n * 6 + 5] - t0));
  }
  return 0;
}''')
open(p, 'w').write(s)
p = os.path.join(ROOT, 'tools/tune/build.sh')
s = open(p).read()
s = s.replace('for v in "4 2 100" "4 3 100" "4 4 100"; do', 'for v in "4 2 100"; do')
open(p, 'w').write(s)
print("ok")
