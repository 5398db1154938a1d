import sys
src, dst = sys.argv[1], sys.argv[2]
s=open(src).read()
s=s.replace('__device__ long long g_ml_prof[16];','__device__ long long g_ml_prof[64];')
s=s.replace('#define RGN_MT(i) if (blockIdx.x == RGN_ML_PROF && threadIdx.x == 0) g_ml_prof[i] = __builtin_readcyclecounter();',
 '#define RGN_MT(i) { __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == RGN_ML_PROF && (threadIdx.x & 255) == 0) g_ml_prof[(i) + 32 * (threadIdx.x >> 8)] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }')
s=s.replace('sizeof(long long) * 16)','sizeof(long long) * 64)')
def rep(a,b):
    global s
    assert s.count(a)==1,(a[:60],s.count(a))
    s=s.replace(a,b)
rep('''        gemm16(acc, a_off_y, g.W1, 32, 16 * c + 2 * wave, 0);    // hidden columns [512 c, 512 c + 512)
''','''        gemm16(acc, a_off_y, g.W1, 32, 16 * c + 2 * wave, 0);    // hidden columns [512 c, 512 c + 512)
        if (c == 0) RGN_MT(7)
        if (c == 1) RGN_MT(11)
''')
rep('''        if (c == 1) __builtin_amdgcn_s_barrier();                     // every wave is done reading the first half's image
        store_img(acc, smem + ML_X);''','''        if (c == 0) RGN_MT(8)
        if (c == 1) __builtin_amdgcn_s_barrier();                     // every wave is done reading the first half's image
        store_img(acc, smem + ML_X);
        if (c == 0) RGN_MT(9)''')
rep('''        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        gemm16(acc2, a_off, g.W2, 16, 2 * wave, 16 * c);       // linear2 over hidden k-blocks [16 c, 16 c + 16)''','''        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (c == 0) RGN_MT(10)
        if (c == 1) RGN_MT(12)
        gemm16(acc2, a_off, g.W2, 16, 2 * wave, 16 * c);       // linear2 over hidden k-blocks [16 c, 16 c + 16)
        if (c == 0) RGN_MT(6)''')
rep('''    add_resid(acc, smem + ML_Y);
    layernorm(acc, vec + V_G1, nullptr, std::false_type{});''','''    add_resid(acc, smem + ML_Y);
    RGN_MT(14)
    layernorm(acc, vec + V_G1, nullptr, std::false_type{});
    RGN_MT(15)''')
rep('''    layernorm(acc, vec + V_G2, vec + V_B2, std::true_type{});
    store_img(acc, smem + ML_Y); ''','''    RGN_MT(16)
    layernorm(acc, vec + V_G2, vec + V_B2, std::true_type{});
    RGN_MT(17)
    store_img(acc, smem + ML_Y); ''')
open(dst,'w').write(s)
