// hipsim.h -- TEST INFRASTRUCTURE ONLY.
//
// A minimal lockstep emulator that lets the HIP kernel sources under
// ganet_amd/csrc/ compile with g++ and run on the CPU, so the CPU test-suite
// (pytest -m "not gpu") can check kernel LOGIC -- indexing, DPP lane patterns,
// segment reductions, LDS tiling, barriers -- against the oracle in a container
// that has no GPU.  It is not a product path: ganet_amd never loads the library
// built from it unless a test injects it explicitly (tests/sim_util.py).
//
// Model: one workgroup at a time; every HIP thread is a ucontext fiber on one OS
// thread; __syncthreads() and the DPP exchange are generation barriers that yield
// to a round-robin scheduler.  Wavefront = 64 consecutive threads.  DPP follows
// the gfx9 ISA: quad_perm, row_shl/shr:n, row_mirror, row_half_mirror within
// 16-lane rows, wave_shl:1 / wave_shr:1 across the wavefront; a lane whose source is out of range keeps `old`.
#pragma once
#include <assert.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <functional>
#include <deque>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };

typedef void *hipStream_t;
typedef void *hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipMemcpyDeviceToDevice = 3, hipEventDisableTiming = 2, hipStreamNonBlocking = 1 };
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline const char *hipGetErrorString(hipError_t) { return "hipsim"; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }

namespace hipsim {

struct Fiber {
  ucontext_t ctx;
  char *stack = nullptr;
  bool done = false;
  bool ext_blocked = false;      // its last yield waited for OTHER wavefronts (workgroup barrier, progress-flag poll)
  dim3 tidx;
};

struct Barrier { int count = 0; int gen = 0; };

struct State {
  ucontext_t sched;
  std::vector<Fiber> fibers;
  int cur = 0;
  int nthreads = 0;
  dim3 blockDim_, gridDim_, blockIdx_;
  Barrier block_bar;
  std::vector<Barrier> wave_bar;
  std::vector<int> xchg;
  std::function<void()> body;
  char *dyn_smem = nullptr;
  // global -> LDS copies (global_load_lds_*) in flight, per lane, oldest first.  late_dma == false: a copy lands when it is
  // issued.  late_dma == true: it lands as LATE as the kernel's own s_waitcnt vmcnt(n) allows -- when the lane executes a
  // wait that leaves at most n copies outstanding -- so a hand-counted wait that is one too loose reads a stale ring slot
  // here, deterministically.  Only the copies are counted: the other vector-memory operations of a wave (tap gathers, result
  // stores) are younger or older than the copies around them and can only make the real wait more conservative than this
  // model (vmcnt retires in issue order on gfx9), never less.
  struct DmaOp { float *dst; float v[4]; int n; };
  std::vector<std::deque<DmaOp>> dma;
  bool late_dma = false;
  long dma_late_landed = 0;
  // LDS reads written in asm (ds_read2_b32 of the planar LGA staging): the compiler does not wait for those, the kernel's own
  // counted s_waitcnt lgkmcnt(n) does.  late_lds: the destination registers hold a signalling pattern until a wait of the lane
  // leaves at most n reads outstanding (the LDS queue of a wave returns in order), i.e. the data arrives as LATE as the
  // kernel's waits permit; consuming a row before its wait, or a count one too loose, computes on NaNs.  Reads the compiler
  // issues itself sit in the same hardware queue and can only make the real wait stricter than this model.
  // *_slack: tests only -- every wait of that counter behaves as if its count were that much larger (a deliberately loosened
  // wait must fail the parity tests).
  struct LdsOp { float *dst; float v[2]; };
  std::vector<std::deque<LdsOp>> lds;
  bool late_lds = false;
  long lds_late_landed = 0;
  int lgkm_slack = 0, vm_slack = 0;
  // order in which the runnable threads of a block are resumed between two barriers: 0 = ascending thread index,
  // 1 = descending.  A hand-off through LDS that lacks a barrier is decided by whichever thread runs first; results that
  // are the same in both orders do not depend on that luck.
  int lane_order = 0;
  // wave_greedy: instead of resuming every thread of the block in turn (which moves all wavefronts forward together), run ONE
  // wavefront for as long as it can go -- until each of its threads waits for another wavefront -- then the next one.  The
  // first wavefront scheduled is then as far ahead of the others as the kernel's synchronisation permits, which is what a
  // hand-off between waves that is too permissive needs in order to show (with lane_order: the last wavefront instead).
  int wave_greedy = 0;
};

inline State &S() { static State s; return s; }

inline void yield(bool waits_for_other_waves = true)
{
  State &s = S();
  s.fibers[s.cur].ext_blocked = waits_for_other_waves;
  swapcontext(&s.fibers[s.cur].ctx, &s.sched);
  s.fibers[s.cur].ext_blocked = false;
}

inline void barrier_wait(Barrier &b, int n, bool across_waves = true)
{
  const int gen = b.gen;
  if (++b.count == n) { b.count = 0; b.gen++; }
  else while (b.gen == gen) yield(across_waves);
}

inline int flat_tid() { State &s = S(); const dim3 &t = s.fibers[s.cur].tidx; return t.x + s.blockDim_.x * (t.y + s.blockDim_.y * t.z); }
inline int lane_id() { return flat_tid() & 63; }
inline void syncthreads() { State &s = S(); barrier_wait(s.block_bar, s.nthreads); }
// hand-off between the lanes of ONE wavefront (GA_WAVE_SYNC): a barrier over the caller's 64 threads only -- in a workgroup of
// several waves it must not synchronise the others, or a missing workgroup barrier would go unnoticed here
inline void wave_sync()
{
  State &s = S();
  const int wave = flat_tid() >> 6;
  const int wsize = (s.nthreads - wave * 64) < 64 ? (s.nthreads - wave * 64) : 64;
  barrier_wait(s.wave_bar[wave], wsize, false);
}

inline void dma_land(const State::DmaOp &op) { for (int k = 0; k < op.n; k++) op.dst[k] = op.v[k]; }
// one lane's share of a global -> LDS copy instruction: n floats from src to dst (n == 0: the lane is masked out of the
// instruction, which the wave's counter counts all the same)
inline void dma_issue(float *dst, const float *src, int n)
{
  State &s = S();
  State::DmaOp op;
  op.dst = dst; op.n = n;
  for (int k = 0; k < n; k++) op.v[k] = src[k];
  if (!s.late_dma) { dma_land(op); return; }
  s.dma[flat_tid()].push_back(op);
}
inline void dma_masked(int count) { for (int i = 0; i < count; i++) dma_issue(nullptr, nullptr, 0); }
// s_waitcnt vmcnt(n)
inline void vmcnt(int n)
{
  State &s = S();
  if (!s.late_dma) return;
  auto &q = s.dma[flat_tid()];
  while ((int)q.size() > n + s.vm_slack) { dma_land(q.front()); q.pop_front(); s.dma_late_landed++; }
}
// one lane's ds_read2_b32 issued from asm: dst[0] = *p0, dst[1] = *p1, visible to the lane after its next covering wait
inline void lds_read2(float *dst, const float *p0, const float *p1)
{
  State &s = S();
  if (!s.late_lds) { dst[0] = *p0; dst[1] = *p1; return; }
  State::LdsOp op;
  op.dst = dst; op.v[0] = *p0; op.v[1] = *p1;      // (sampled at issue: the ring slot is not rewritten while reads of it are in flight)
  dst[0] = dst[1] = __builtin_nanf("0x5152");
  s.lds[flat_tid()].push_back(op);
}
// s_waitcnt lgkmcnt(n)
inline void lgkmcnt(int n)
{
  State &s = S();
  if (!s.late_lds) return;
  auto &q = s.lds[flat_tid()];
  while ((int)q.size() > n + s.lgkm_slack) { q.front().dst[0] = q.front().v[0]; q.front().dst[1] = q.front().v[1]; q.pop_front(); s.lds_late_landed++; }
}

inline int update_dpp(int old, int src, int ctrl)
{
  State &s = S();
  const int tid = flat_tid(), wave = tid >> 6, lane = tid & 63;
  const int wsize = (s.nthreads - wave * 64) < 64 ? (s.nthreads - wave * 64) : 64;
  s.xchg[tid] = src;
  barrier_wait(s.wave_bar[wave], wsize, false);
  int sl = lane;
  bool valid = true;
  if (ctrl >= 0 && ctrl <= 0xFF) sl = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  else if (ctrl >= 0x101 && ctrl <= 0x10F) { const int r = (lane & 15) + (ctrl - 0x100); valid = r < 16; sl = (lane & ~15) + r; }
  else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int r = (lane & 15) - (ctrl - 0x110); valid = r >= 0; sl = (lane & ~15) + r; }
  else if (ctrl == 0x130) { valid = lane < 63; sl = lane + 1; }     // wave_shl:1
  else if (ctrl == 0x138) { valid = lane > 0; sl = lane - 1; }      // wave_shr:1
  else if (ctrl == 0x140) sl = (lane & ~15) + (15 - (lane & 15));
  else if (ctrl == 0x141) sl = (lane & ~7) + (7 - (lane & 7));
  else { fprintf(stderr, "hipsim: unsupported dpp_ctrl 0x%x\n", ctrl); abort(); }
  if (valid && sl >= wsize) valid = false;   // inactive source lane: dest keeps old
  const int v = valid ? s.xchg[wave * 64 + sl] : old;
  barrier_wait(s.wave_bar[wave], wsize, false);
  return v;
}

// v_readlane_b32: the value lane `l` of the caller's wavefront holds
inline int readlane(int src, int l)
{
  State &s = S();
  const int tid = flat_tid(), wave = tid >> 6;
  const int wsize = (s.nthreads - wave * 64) < 64 ? (s.nthreads - wave * 64) : 64;
  s.xchg[tid] = src;
  barrier_wait(s.wave_bar[wave], wsize, false);
  const int v = s.xchg[wave * 64 + (l < wsize ? l : 0)];
  barrier_wait(s.wave_bar[wave], wsize, false);
  return v;
}

inline void fiber_entry()
{
  State &s = S();
  s.body();
  if (s.late_dma && s.vm_slack > 0) {              // (loosened on purpose: the final wait leaves copies behind; let them land)
    auto &q = s.dma[flat_tid()];
    while (!q.empty()) { dma_land(q.front()); q.pop_front(); }
  }
  if (s.late_dma && !s.dma[flat_tid()].empty()) {
    fprintf(stderr, "hipsim: thread %d ended with %d global->LDS copies in flight (its LDS may already belong to the next workgroup)\n",
            flat_tid(), (int)s.dma[flat_tid()].size());
    abort();
  }
  if (s.late_lds) s.lds[flat_tid()].clear();      // (registers die with the thread)
  s.fibers[s.cur].done = true;
  swapcontext(&s.fibers[s.cur].ctx, &s.sched);
}

inline void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> body)
{
  State &s = S();
  const int nt = (int)(block.x * block.y * block.z);
  const size_t STACK = 256 * 1024;
  s.nthreads = nt;
  s.blockDim_ = block;
  s.gridDim_ = grid;
  s.body = body;
  if ((int)s.fibers.size() < nt) {
    const size_t old = s.fibers.size();
    s.fibers.resize(nt);
    for (size_t i = old; i < (size_t)nt; i++) s.fibers[i].stack = (char *)malloc(STACK);
  }
  s.wave_bar.assign((nt + 63) / 64, Barrier());
  s.xchg.assign(nt, 0);
  s.dma.assign(nt, std::deque<State::DmaOp>());
  s.lds.assign(nt, std::deque<State::LdsOp>());
  std::vector<char> smem(shmem + 64);
  s.dyn_smem = (char *)(((uintptr_t)smem.data() + 15) & ~(uintptr_t)15);
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
        s.blockIdx_ = dim3(bx, by, bz);
        s.block_bar = Barrier();
        for (auto &w : s.wave_bar) w = Barrier();
        for (int t = 0; t < nt; t++) {
          Fiber &f = s.fibers[t];
          f.done = false;
          f.ext_blocked = false;
          f.tidx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = f.stack;
          f.ctx.uc_stack.ss_size = STACK;
          f.ctx.uc_link = &s.sched;
          makecontext(&f.ctx, (void (*)())fiber_entry, 0);
        }
        int remaining = nt;
        long spins = 0;
        const int nwave = (nt + 63) / 64;
        while (remaining > 0) {
          if (s.wave_greedy && nwave > 1) {
            for (int wi = 0; wi < nwave; wi++) {
              const int w = s.lane_order ? nwave - 1 - wi : wi;
              const int t0 = w * 64, t1 = t0 + 64 < nt ? t0 + 64 : nt;
              for (;;) {                               // this wavefront, until all of its threads wait for another one
                bool can_go_on = false;
                for (int i = t0; i < t1; i++) {
                  const int t = s.lane_order ? t1 - 1 - (i - t0) : i;
                  if (s.fibers[t].done) continue;
                  s.cur = t;
                  swapcontext(&s.sched, &s.fibers[t].ctx);
                  if (s.fibers[t].done) remaining--;
                  else if (!s.fibers[t].ext_blocked) can_go_on = true;
                }
                if (!can_go_on) break;
                if (++spins > 100000000L) { fprintf(stderr, "hipsim: deadlock inside a wavefront (divergent wave barrier/DPP?)\n"); abort(); }
              }
            }
          } else {
            for (int i = 0; i < nt; i++) {
              const int t = s.lane_order ? nt - 1 - i : i;
              if (s.fibers[t].done) continue;
              s.cur = t;
              swapcontext(&s.sched, &s.fibers[t].ctx);
              if (s.fibers[t].done) remaining--;
            }
          }
          if (++spins > 100000000L) { fprintf(stderr, "hipsim: deadlock (divergent barrier/DPP?)\n"); abort(); }
        }
      }
}

}  // namespace hipsim

#define threadIdx (hipsim::S().fibers[hipsim::S().cur].tidx)
#define blockIdx (hipsim::S().blockIdx_)
#define blockDim (hipsim::S().blockDim_)
#define gridDim (hipsim::S().gridDim_)
#define __syncthreads() hipsim::syncthreads()
