// TEST TOOLING ONLY -- runtime of the host emulation of HIP workgroups (see hip/hip_runtime.h).
//
// One workgroup at a time per OS thread; its lanes are stackful coroutines switched by a tiny
// x86-64 context switch.  Rendezvous primitives:
//   * block barrier   (__syncthreads)
//   * wave rendezvous (MFMA 16x16x4 / 32x32x2 f32, shuffles, ballot): lanes deposit operands into a
//     generation-parity double buffer, meet, then each lane computes its own outputs.
// MFMA numerics follow the hardware: D = fma(a_k3,b_k3, fma(a_k2,b_k2, fma(a_k1,b_k1, fma(a_k0,b_k0, C))))
// (k-ordered fmaf chain, one rounding per product) -- cdna_hip_programming.md §3.
#include <hip/hip_runtime.h>
#undef threadIdx
#undef blockIdx
#undef blockDim
#undef gridDim

#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

namespace emu {

// ------------------------------------------------------------------ context switch
struct Ctx { void* sp; };
extern "C" void emu_switch(Ctx* from, Ctx* to);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq  $8, %rsp
    stmxcsr (%rsp)
    fnstcw  4(%rsp)
    movq  %rsp, (%rdi)
    movq  (%rsi), %rsp
    ldmxcsr (%rsp)
    fldcw   4(%rsp)
    addq  $8, %rsp
    popq  %r15
    popq  %r14
    popq  %r13
    popq  %r12
    popq  %rbx
    popq  %rbp
    ret
.size emu_switch,.-emu_switch
)");

struct WaveScratch {
    float a[2][64];
    float b[2][64];
    unsigned short ha[2][64][8], hb[2][64][8];      // bf16 fragments (v_mfma_f32_16x16x32_bf16)
    int ia[2][64];
    int arrived = 0;
    unsigned gen = 0;
    int live = 0;      // lanes of this wave that have not returned
};

struct Lane {
    Ctx ctx{};
    char* stack = nullptr;
    dim3 tid;
    int linear = 0;
    int lane = 0;      // lane within wave
    int wave = 0;
    bool done = false;
};

struct Worker {
    BlockCtx blk;
    std::vector<Lane> lanes;
    std::vector<WaveScratch> waves;
    Ctx sched{};
    Lane* cur = nullptr;
    int bar_arrived = 0;
    unsigned bar_gen = 0;
    int live = 0;
    const std::function<void()>* body = nullptr;
    std::vector<char*> stack_pool;
};

static thread_local Worker* tw = nullptr;
static const size_t STACK_BYTES = 256 * 1024;

Lane* cur() { return tw->cur; }
BlockCtx& blk() { return tw->blk; }
const dim3& tid() { return tw->cur->tid; }
int lane_id() { return tw->cur->lane; }

static void yield() { emu_switch(&tw->cur->ctx, &tw->sched); }

static void lane_entry() {
    Worker* w = tw;
    (*w->body)();
    Lane* l = w->cur;
    l->done = true;
    w->live--;
    w->waves[l->wave].live--;
    // a finished lane counts as arrived at every later rendezvous: release waiters if it was the last
    if (w->bar_arrived > 0 && w->bar_arrived >= w->live) { w->bar_arrived = 0; w->bar_gen++; }
    WaveScratch& ws = w->waves[l->wave];
    if (ws.arrived > 0 && ws.arrived >= ws.live) { ws.arrived = 0; ws.gen++; }
    emu_switch(&l->ctx, &w->sched);
    fprintf(stderr, "emu: resumed a finished lane\n");
    abort();
}

void syncthreads() {
    Worker* w = tw;
    unsigned g = w->bar_gen;
    if (++w->bar_arrived >= w->live) { w->bar_arrived = 0; w->bar_gen++; return; }
    while (w->bar_gen == g) yield();
}

static unsigned wave_rendezvous(WaveScratch& ws) {
    unsigned g = ws.gen;
    if (++ws.arrived >= ws.live) { ws.arrived = 0; ws.gen++; return g; }
    while (ws.gen == g) yield();
    return g;
}

float shfl_f(float v, int src, int width) {
    Worker* w = tw; Lane* l = w->cur; WaveScratch& ws = w->waves[l->wave];
    int p = ws.gen & 1;
    ws.a[p][l->lane] = v;
    wave_rendezvous(ws);
    if (width < 64) src = (l->lane / width) * width + (src % width);
    return ws.a[p][src & 63];
}
int shfl_i(int v, int src, int width) {
    Worker* w = tw; Lane* l = w->cur; WaveScratch& ws = w->waves[l->wave];
    int p = ws.gen & 1;
    ws.ia[p][l->lane] = v;
    wave_rendezvous(ws);
    if (width < 64) src = (l->lane / width) * width + (src % width);
    return ws.ia[p][src & 63];
}
unsigned long long ballot(int pred) {
    Worker* w = tw; Lane* l = w->cur; WaveScratch& ws = w->waves[l->wave];
    int p = ws.gen & 1;
    ws.ia[p][l->lane] = pred ? 1 : 0;
    int nl = ws.live;
    wave_rendezvous(ws);
    unsigned long long m = 0;
    for (int i = 0; i < 64; ++i) if (i < 64 && ws.ia[p][i] && i < (int)(w->lanes.size() - l->wave * 64)) m |= 1ull << i;
    (void)nl;
    return m;
}

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: col=l&15, row=(l>>4)*4+r
void mfma16(float a, float b, float* c4) {
    Worker* w = tw; Lane* l = w->cur; WaveScratch& ws = w->waves[l->wave];
    int p = ws.gen & 1;
    ws.a[p][l->lane] = a;
    ws.b[p][l->lane] = b;
    wave_rendezvous(ws);
    int col = l->lane & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l->lane >> 4) * 4 + r;
        float acc = c4[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(ws.a[p][k * 16 + row], ws.b[p][k * 16 + col], acc);
        c4[r] = acc;
    }
}
// v_mfma_f32_4x4x1_16B_f32 (layout verified on gfx950 by tools/ubench/mfma_4x4_layout.cpp): sixteen 4x4 blocks, block = lane >> 2; lane (b, i) supplies
// A_b[i] and B_b[i]; register r of lane (b, j) accumulates A_b[r] * B_b[j]
void mfma4(float a, float b, float* c4) {
    Worker* w = tw; Lane* l = w->cur; WaveScratch& ws = w->waves[l->wave];
    int p = ws.gen & 1;
    ws.a[p][l->lane] = a;
    ws.b[p][l->lane] = b;
    wave_rendezvous(ws);
    const int blk = l->lane >> 2;
    for (int r = 0; r < 4; ++r) c4[r] = fmaf(ws.a[p][4 * blk + r], ws.b[p][l->lane], c4[r]);
}
// v_mfma_f32_16x16x32_bf16 (layout verified on gfx950 by tools/ubench/mfma_bf16_layout.cpp): lane l holds A[i=l&15][k=8(l>>4)..+7],
// B[k=8(l>>4)..+7][j=l&15]; D as the 16x16x4 form.  Products of bf16 values are exact in f32; the accumulation order over k is the
// hardware's business -- emulated as one f32 sum in ascending k (the kernels' tolerance covers either).
void mfma16_bf16(const unsigned short* a8, const unsigned short* b8, float* c4) {
    Worker* w = tw; Lane* l = w->cur; WaveScratch& ws = w->waves[l->wave];
    int p = ws.gen & 1;
    for (int q = 0; q < 8; ++q) { ws.ha[p][l->lane][q] = a8[q]; ws.hb[p][l->lane][q] = b8[q]; }
    wave_rendezvous(ws);
    auto f = [](unsigned short h) { unsigned u = (unsigned)h << 16; float x; memcpy(&x, &u, 4); return x; };
    int col = l->lane & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l->lane >> 4) * 4 + r;
        float acc = c4[r];
        for (int k = 0; k < 32; ++k) acc += f(ws.ha[p][(k >> 3) * 16 + row][k & 7]) * f(ws.hb[p][(k >> 3) * 16 + col][k & 7]);
        c4[r] = acc;
    }
}
void mfma16_f16(const unsigned short* a8, const unsigned short* b8, float* c4) {
    Worker* w = tw; Lane* l = w->cur; WaveScratch& ws = w->waves[l->wave];
    int p = ws.gen & 1;
    for (int q = 0; q < 8; ++q) { ws.ha[p][l->lane][q] = a8[q]; ws.hb[p][l->lane][q] = b8[q]; }
    wave_rendezvous(ws);
    auto f = [](unsigned short h) { _Float16 x; memcpy(&x, &h, 2); return (float)x; };
    int col = l->lane & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l->lane >> 4) * 4 + r;
        float acc = c4[r];
        for (int k = 0; k < 32; ++k) acc += f(ws.ha[p][(k >> 3) * 16 + row][k & 7]) * f(ws.hb[p][(k >> 3) * 16 + col][k & 7]);
        c4[r] = acc;
    }
}
// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
void mfma32(float a, float b, float* c16) {
    Worker* w = tw; Lane* l = w->cur; WaveScratch& ws = w->waves[l->wave];
    int p = ws.gen & 1;
    ws.a[p][l->lane] = a;
    ws.b[p][l->lane] = b;
    wave_rendezvous(ws);
    int col = l->lane & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l->lane >> 5);
        float acc = c16[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(ws.a[p][k * 32 + row], ws.b[p][k * 32 + col], acc);
        c16[r] = acc;
    }
}

static void run_block(Worker* w, dim3 bidx, dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    tw = w;
    int nthreads = (int)(block.x * block.y * block.z);
    int nwaves = (nthreads + 63) / 64;
    w->blk.blockIdx = bidx; w->blk.blockDim = block; w->blk.gridDim = grid;
    if (w->blk.lds_bytes < shmem + 64) {
        free(w->blk.lds);
        w->blk.lds = (unsigned char*)aligned_alloc(64, ((shmem + 64 + 63) / 64) * 64);
        w->blk.lds_bytes = shmem + 64;
    }
    // poison LDS so that reads of never-written words show up as NaNs
    memset(w->blk.lds, 0xFF, shmem);
    w->lanes.assign(nthreads, Lane());
    w->waves.assign(nwaves, WaveScratch());
    while ((int)w->stack_pool.size() < nthreads) w->stack_pool.push_back((char*)aligned_alloc(64, STACK_BYTES));
    w->bar_arrived = 0; w->bar_gen = 0; w->live = nthreads; w->body = &body;
    for (int t = 0; t < nthreads; ++t) {
        Lane& l = w->lanes[t];
        l.linear = t;
        l.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        l.lane = t & 63; l.wave = t >> 6; l.done = false;
        w->waves[l.wave].live++;
        l.stack = w->stack_pool[t];
        // initial frame consumed by emu_switch: [mxcsr|fpcw pad 8][r15 r14 r13 r12 rbx rbp][ret addr]
        uintptr_t top = ((uintptr_t)l.stack + STACK_BYTES) & ~(uintptr_t)63;
        void** sp = (void**)top;
        *--sp = nullptr;                 // fake return address for lane_entry (keeps 16B alignment at entry)
        *--sp = (void*)&lane_entry;      // ret target
        for (int i = 0; i < 6; ++i) *--sp = nullptr;
        --sp;
        uint32_t mx = 0x1F80; uint16_t cw = 0x037F;
        memcpy((char*)sp, &mx, 4); memcpy((char*)sp + 4, &cw, 2);
        l.ctx.sp = sp;
    }
    int remaining = nthreads;
    while (remaining > 0) {
        remaining = 0;
        for (int t = 0; t < nthreads; ++t) {
            Lane& l = w->lanes[t];
            if (l.done) continue;
            w->cur = &l;
            emu_switch(&w->sched, &l.ctx);
            if (!l.done) remaining++;
        }
    }
    w->cur = nullptr;
}

static int n_workers() {
    const char* e = getenv("FM_EMU_THREADS");
    int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    return n < 1 ? 1 : (n > 64 ? 64 : n);
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    size_t nblocks = (size_t)grid.x * grid.y * grid.z;
    if (nblocks == 0) return;
    int nw = (int)std::min<size_t>(n_workers(), nblocks);
    std::atomic<size_t> next{0};
    auto work = [&]() {
        Worker* w = new Worker();
        w->blk.lds = nullptr; w->blk.lds_bytes = 0;
        for (;;) {
            size_t b = next.fetch_add(1);
            if (b >= nblocks) break;
            dim3 bidx((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((size_t)grid.x * grid.y)));
            run_block(w, bidx, grid, block, shmem, body);
        }
        for (char* s : w->stack_pool) free(s);
        free(w->blk.lds);
        delete w;
        tw = nullptr;
    };
    if (nw == 1) { work(); return; }
    std::vector<std::thread> th;
    for (int i = 0; i < nw; ++i) th.emplace_back(work);
    for (auto& t : th) t.join();
}

}  // namespace emu

double emu_now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}
