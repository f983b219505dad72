// Device-resident graph-map state of a batch of episodes and the assembly of the navigation inputs from it
// (SURVEY 8f row f1).  Replaces the Python side of a rollout step: GraphMap / FloydGraph
// (finetune_src/models/graph_utils.py:42-148), the per-step graph and node-embedding updates
// (finetune_src/r2r/agent.py:406-408, 434-451, 565-569) and the two padding builders _nav_gmap_variable /
// _nav_vp_variable (agent.py:157-278) - dict-of-dict Floyd updates, O(G^2) Python loops, per-node torch.stack and ~8
// host-to-device copies per step - by three launches over state that never leaves the GPU.
// One workgroup per episode.  Distances are kept in double and compared exactly like the reference's Python floats
// (no FMA contraction in this file), so the graph decisions (edge relaxations, via-points, hop counts) are identical.
#include "common.h"
#pragma clang fp contract(off)

namespace {

constexpr double INF_DIST = 95959595.0;   // graph_utils.py:44
constexpr int MAXN = 128;                  // node slots per episode supported by the hop-count stack

struct GraphState {
  double* pos;        // [B,N,3]
  double* dist;       // [B,N,N]   INF_DIST = unknown (diagonal included, like the reference's defaultdict)
  int32_t* point;     // [B,N,N]   via-node of the shortest path, -1 = direct edge / unknown
  uint8_t* visited;   // [B,N]
  int32_t* step_id;   // [B,N]
  float* emb_sum;     // [B,N,H]
  int32_t* emb_cnt;   // [B,N]
  int32_t* n_nodes;   // [B]
  int N, H;
};

__global__ __launch_bounds__(256) void graph_fill_kernel(double* __restrict__ d, int32_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { d[i] = INF_DIST; p[i] = -1; }
}

// ---- update_graph(ob) + node_step_ids[cur] = t+1  (graph_utils.py:106-112, 55-72; agent.py:434-437)
__global__ __launch_bounds__(256) void graph_observe_kernel(GraphState s, const uint8_t* __restrict__ graph_active,
                                                            const uint8_t* __restrict__ step_active, const int32_t* __restrict__ cur_slot,
                                                            const int32_t* __restrict__ n_new, const int32_t* __restrict__ ncand,
                                                            const int32_t* __restrict__ cand_slot, const double* __restrict__ cur_pos,
                                                            const double* __restrict__ cand_pos, int Cmax, int step_value) {
  const int b = blockIdx.x, tid = threadIdx.x, N = s.N;
  const int k = cur_slot[b];
  if (graph_active[b]) {
    double* pos = s.pos + (size_t)b * N * 3;
    double* dist = s.dist + (size_t)b * N * N;
    int32_t* point = s.point + (size_t)b * N * N;
    const int nc = ncand[b], n = n_new[b];
    if (tid < 3) pos[k * 3 + tid] = cur_pos[b * 3 + tid];
    if (tid < nc * 3) pos[cand_slot[b * Cmax + tid / 3] * 3 + tid % 3] = cand_pos[((size_t)b * Cmax + tid / 3) * 3 + tid % 3];
    if (tid < nc) {                                   // add_edge(cur, cand, |cur - cand|)
      const int c = cand_slot[b * Cmax + tid];
      const double dx = cand_pos[((size_t)b * Cmax + tid) * 3] - cur_pos[b * 3];
      const double dy = cand_pos[((size_t)b * Cmax + tid) * 3 + 1] - cur_pos[b * 3 + 1];
      const double dz = cand_pos[((size_t)b * Cmax + tid) * 3 + 2] - cur_pos[b * 3 + 2];
      const double d = sqrt(dx * dx + dy * dy + dz * dz);
      if (d < dist[k * N + c]) {
        dist[k * N + c] = dist[c * N + k] = d;
        point[k * N + c] = point[c * N + k] = -1;
      }
    }
    if (tid == 0) { s.n_nodes[b] = n; }
    __syncthreads();
    // graph.update(k): relax every pair through k.  Row / column k cannot change here (it would need dist[k][k]).
    for (int p = tid; p < n * n; p += 256) {
      const int x = p / n, y = p - x * n;
      if (x < y) {
        const double via = dist[x * N + k] + dist[k * N + y];
        if (via < dist[x * N + y]) {
          dist[x * N + y] = dist[y * N + x] = via;
          point[x * N + y] = point[y * N + x] = k;
        }
      }
    }
    if (tid == 0) s.visited[(size_t)b * N + k] = 1;
  }
  if (step_active[b] && tid == 0) s.step_id[(size_t)b * N + k] = step_value;
}

// ---- node embeddings (agent.py:443-451): the current node is rewritten with the GAA-pooled panorama, every unvisited
// candidate accumulates the embedding of the view it was seen through
template <typename T>
__global__ __launch_bounds__(256) void graph_embed_kernel(GraphState s, const uint8_t* __restrict__ active, const int32_t* __restrict__ cur_slot,
                                                          const int32_t* __restrict__ ncand, const int32_t* __restrict__ cand_slot, int Cmax,
                                                          const T* __restrict__ pano, int64_t pano_bs, int pano_rs,
                                                          const T* __restrict__ fused, int fused_rs) {
  const int b = blockIdx.x, N = s.N, H = s.H;
  if (!active[b]) return;
  const int k = cur_slot[b], nc = ncand[b];
  float* es = s.emb_sum + (size_t)b * N * H;
  for (int c = threadIdx.x * 4; c < H; c += 1024) *reinterpret_cast<float4*>(es + (size_t)k * H + c) = ld4(fused + (size_t)b * fused_rs + c);
  for (int j = 0; j < nc; ++j) {
    const int slot = cand_slot[b * Cmax + j];
    if (s.visited[(size_t)b * N + slot]) continue;
    const bool first = s.emb_cnt[(size_t)b * N + slot] == 0;
    for (int c = threadIdx.x * 4; c < H; c += 1024) {
      float4 v = ld4(pano + b * pano_bs + (size_t)j * pano_rs + c);
      if (!first) {
        const float4 o = *reinterpret_cast<const float4*>(es + (size_t)slot * H + c);
        v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
      }
      *reinterpret_cast<float4*>(es + (size_t)slot * H + c) = v;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s.emb_cnt[(size_t)b * N + k] = 1;
    for (int j = 0; j < nc; ++j) {
      const int slot = cand_slot[b * Cmax + j];
      if (!s.visited[(size_t)b * N + slot]) s.emb_cnt[(size_t)b * N + slot] += 1;
    }
  }
}

// len(graph.path(x, y)) (graph_utils.py:77-93): the via-points are followed lazily, exactly like the recursion
__device__ int path_hops(const int32_t* __restrict__ point, int N, int x, int y) {
  short sx[MAXN], sy[MAXN];
  int top = 0, hops = 0;
  sx[0] = (short)x; sy[0] = (short)y; top = 1;
  while (top > 0) {
    --top;
    const int a = sx[top], c = sy[top];
    if (a == c) continue;
    const int k = point[a * N + c];
    if (k < 0) { ++hops; continue; }
    if (top + 2 > MAXN) return hops;       // cannot happen for N <= MAXN
    sx[top] = (short)k; sy[top] = (short)c; ++top;
    sx[top] = (short)a; sy[top] = (short)k; ++top;
  }
  return hops;
}

// get_pos_fts row (graph_utils.py:126-148 with :15-40): 7 floats for node `slot` seen from `cur`; slot < 0 = the None token
__device__ void pos_fts_row(const double* __restrict__ pos, const double* __restrict__ dist, const int32_t* __restrict__ point, int N, int cur,
                            int slot, double base_heading, double base_elevation, float* __restrict__ out) {
  if (slot < 0) {
    out[0] = 0.f; out[1] = 1.f; out[2] = 0.f; out[3] = 1.f; out[4] = 0.f; out[5] = 0.f; out[6] = 0.f;
    return;
  }
  const double dx = pos[slot * 3] - pos[cur * 3], dy = pos[slot * 3 + 1] - pos[cur * 3 + 1], dz = pos[slot * 3 + 2] - pos[cur * 3 + 2];
  const double xy = fmax(sqrt(dx * dx + dy * dy), 1e-8), xyz = fmax(sqrt(dx * dx + dy * dy + dz * dz), 1e-8);
  double heading = asin(dx / xy);
  if (pos[slot * 3 + 1] < pos[cur * 3 + 1]) heading = 3.141592653589793 - heading;
  heading -= base_heading;
  const double elevation = asin(dz / xyz) - base_elevation;
  const float hf = (float)heading, ef = (float)elevation;      // .astype(np.float32) BEFORE the float32 sin / cos
  out[0] = sinf(hf); out[1] = cosf(hf); out[2] = sinf(ef); out[3] = cosf(ef);
  const double gd = slot == cur ? 0.0 : dist[cur * N + slot];
  out[4] = (float)(xyz / 30.0);
  out[5] = (float)(gd / 30.0);
  out[6] = (float)((double)path_hops(point, N, cur, slot) / 10.0);
}

struct NavOut {
  void* gmap_img;          // [B,G,H] storage dtype
  int64_t* gmap_step_ids;  // [B,G]
  float* gmap_pos_fts;     // [B,G,7]
  float* gmap_pair_dists;  // [B,G,G]
  uint8_t* gmap_masks;     // [B,G]
  uint8_t* gmap_visited;   // [B,G]
  int32_t* gmap_slots;     // [B,G]  node slot of each token (-1 for [stop], [MEM], padding)
  int32_t* gmap_src;       // [B,G]  logit-fusion source (vilmodel.py:790-805), see dsrg_nav_logits_fwd
  uint8_t* bw_mask;        // [B,Vp]
  float* vp_pos_fts;       // [B,Vp,14]
  uint8_t* no_vp_left;     // [B]
  int32_t* gmap_len;       // [B]
  int G, Vp;
};

template <typename T>
__global__ __launch_bounds__(256) void graph_nav_inputs_kernel(GraphState s, NavOut o, const int32_t* __restrict__ cur_slot,
                                                               const double* __restrict__ heading, const double* __restrict__ elevation,
                                                               const int32_t* __restrict__ ncand, const int32_t* __restrict__ cand_slot, int Cmax,
                                                               const T* __restrict__ mem, int mem_rs) {
  __shared__ int tok[MAXN + 2];
  __shared__ int len_s;
  const int b = blockIdx.x, tid = threadIdx.x, N = s.N, H = s.H, G = o.G, Vp = o.Vp;
  const int n = s.n_nodes[b], cur = cur_slot[b], nc = ncand[b];
  const double* pos = s.pos + (size_t)b * N * 3;
  const double* dist = s.dist + (size_t)b * N * N;
  const int32_t* point = s.point + (size_t)b * N * N;
  const uint8_t* vis = s.visited + (size_t)b * N;
  if (tid == 0) {            // stop -> memory -> visited -> unvisited, each group in insertion (slot) order (agent.py:165-181)
    int l = 2, unv = 0;
    tok[0] = -1; tok[1] = -1;
    for (int i = 0; i < n; ++i) if (vis[i]) tok[l++] = i;
    for (int i = 0; i < n; ++i) if (!vis[i]) { tok[l++] = i; ++unv; }
    len_s = l;
    o.no_vp_left[b] = unv == 0;
    o.gmap_len[b] = l;
  }
  __syncthreads();
  const int len = min(len_s, G);
  const double bh = heading[b], be = elevation[b];
  for (int j = tid; j < G; j += 256) {
    const size_t r = (size_t)b * G + j;
    const int slot = j < len ? tok[j] : -1;
    o.gmap_slots[r] = slot;
    o.gmap_step_ids[r] = slot >= 0 ? s.step_id[(size_t)b * N + slot] : 0;
    o.gmap_visited[r] = slot >= 0 && vis[slot];
    o.gmap_masks[r] = j < len && j != 1;
    if (j < len) pos_fts_row(pos, dist, point, N, cur, slot, bh, be, o.gmap_pos_fts + r * 7);
    else for (int e = 0; e < 7; ++e) o.gmap_pos_fts[r * 7 + e] = 0.f;
    // logit fusion indices: an unvisited node takes the local score of the candidate that shows it, else the back-track sum
    int src = -1;
    if (slot >= 0 && !vis[slot]) {
      src = -2;
      for (int c = 0; c < nc; ++c) if (cand_slot[b * Cmax + c] == slot) src = c + 2;
    }
    o.gmap_src[r] = src;
  }
  for (int p = tid; p < G * G; p += 256) {
    const int a = p / G, c = p - a * G;
    float v = 0.f;
    if (a >= 2 && c >= 2 && a < len && c < len && a != c) v = (float)dist[tok[a] * N + tok[c]];
    o.gmap_pair_dists[(size_t)b * G * G + p] = v;
  }
  // viewpoint tokens: [:, :7] = the start node seen from here (every row, padding included), [2 : 2+ncand, 7:] = the candidates
  __shared__ float start_fts[7];
  if (tid == 0) pos_fts_row(pos, dist, point, N, cur, 0, bh, be, start_fts);
  __syncthreads();
  for (int j = tid; j < Vp; j += 256) {
    float* row = o.vp_pos_fts + ((size_t)b * Vp + j) * 14;
    for (int e = 0; e < 7; ++e) row[e] = start_fts[e];
    if (j >= 2 && j < 2 + nc) pos_fts_row(pos, dist, point, N, cur, cand_slot[b * Cmax + j - 2], bh, be, row + 7);
    else for (int e = 0; e < 7; ++e) row[7 + e] = 0.f;
    o.bw_mask[(size_t)b * Vp + j] = (j >= 2 && j < 2 + nc) ? vis[cand_slot[b * Cmax + j - 2]] : 0;
  }
  // graph-map token embeddings: zeros, [MEM], mean node embeddings, zero padding
  T* gi = reinterpret_cast<T*>(o.gmap_img) + (size_t)b * G * H;
  for (int p = tid; p < G * (H / 4); p += 256) {
    const int j = p / (H / 4), c = (p - j * (H / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j == 1 && mem) v = ld4(mem + (size_t)b * mem_rs + c);
    else if (j >= 2 && j < len) {
      const int slot = tok[j];
      const float cnt = (float)s.emb_cnt[(size_t)b * N + slot];
      const float4 e = *reinterpret_cast<const float4*>(s.emb_sum + ((size_t)b * N + slot) * H + c);
      v = make_float4(e.x / cnt, e.y / cnt, e.z / cnt, e.w / cnt);
    }
    st4(gi + (size_t)j * H + c, v);
  }
}

}  // namespace

extern "C" int dsrg_graph_reset(double* pos, double* dist, int32_t* point, uint8_t* visited, int32_t* step_id, float* emb_sum,
                                int32_t* emb_cnt, int32_t* n_nodes, int B, int N, int H, void* stream) {
  DSRG_CHECK(pos && dist && point && visited && step_id && emb_sum && emb_cnt && n_nodes && B >= 0 && N >= 1 && N <= MAXN && H >= 4 && H % 4 == 0,
             "dsrg_graph_reset: bad arguments (N <= %d)", MAXN);
  hipStream_t s = (hipStream_t)stream;
  const size_t nn = (size_t)B * N * N;        // dist <- "unknown", via-points <- none
  if (nn) hipLaunchKernelGGL(graph_fill_kernel, dim3((unsigned)((nn + 255) / 256 < 1024 ? (nn + 255) / 256 : 1024)), dim3(256), 0, s, dist, point, nn);
  (void)hipMemsetAsync(pos, 0, (size_t)B * N * 3 * sizeof(double), s);
  (void)hipMemsetAsync(visited, 0, (size_t)B * N, s);
  (void)hipMemsetAsync(step_id, 0, (size_t)B * N * 4, s);
  (void)hipMemsetAsync(emb_sum, 0, (size_t)B * N * H * 4, s);
  (void)hipMemsetAsync(emb_cnt, 0, (size_t)B * N * 4, s);
  (void)hipMemsetAsync(n_nodes, 0, (size_t)B * 4, s);
  DSRG_LAUNCH_CHECK("dsrg_graph_reset");
  return DSRG_OK;
}

extern "C" int dsrg_graph_observe(const dsrg_graph_state_t* st, const uint8_t* graph_active, const uint8_t* step_active,
                                  const int32_t* cur_slot, const int32_t* n_nodes_new, const int32_t* ncand, const int32_t* cand_slot,
                                  const double* cur_pos, const double* cand_pos, int Cmax, int step_value, int B, void* stream) {
  DSRG_CHECK(st && graph_active && step_active && cur_slot && n_nodes_new && ncand && cand_slot && cur_pos && cand_pos, "dsrg_graph_observe: null pointer");
  DSRG_CHECK(st->N >= 1 && st->N <= MAXN && Cmax >= 1 && Cmax * 3 <= 256 && B >= 0, "dsrg_graph_observe: need N <= %d and Cmax <= 85", MAXN);
  if (B == 0) return DSRG_OK;
  GraphState s{st->pos, st->dist, st->point, st->visited, st->step_id, st->emb_sum, st->emb_cnt, st->n_nodes, st->N, st->H};
  hipLaunchKernelGGL(graph_observe_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, s, graph_active, step_active, cur_slot, n_nodes_new, ncand,
                     cand_slot, cur_pos, cand_pos, Cmax, step_value);
  DSRG_LAUNCH_CHECK("dsrg_graph_observe");
  return DSRG_OK;
}

extern "C" int dsrg_graph_update_embeds(const dsrg_graph_state_t* st, const uint8_t* active, const int32_t* cur_slot, const int32_t* ncand,
                                        const int32_t* cand_slot, int Cmax, const void* pano, int64_t pano_bs, int pano_rs,
                                        const void* fused, int fused_rs, int B, int dtype, void* stream) {
  DSRG_CHECK(st && active && cur_slot && ncand && cand_slot && pano && fused, "dsrg_graph_update_embeds: null pointer");
  DSRG_CHECK(dtype == DSRG_F32 || dtype == DSRG_BF16, "dsrg_graph_update_embeds: bad dtype");
  DSRG_CHECK(st->H % 4 == 0 && pano_rs % 4 == 0 && pano_bs % 4 == 0 && fused_rs % 4 == 0 && pano_rs >= st->H && fused_rs >= st->H,
             "dsrg_graph_update_embeds: strides must be >= H and multiples of 4");
  const uintptr_t m = dtype == DSRG_BF16 ? 7 : 15;
  DSRG_CHECK((((uintptr_t)pano | (uintptr_t)fused) & m) == 0, "dsrg_graph_update_embeds: misaligned pointer");
  if (B == 0) return DSRG_OK;
  GraphState s{st->pos, st->dist, st->point, st->visited, st->step_id, st->emb_sum, st->emb_cnt, st->n_nodes, st->N, st->H};
  if (dtype == DSRG_F32)
    hipLaunchKernelGGL((graph_embed_kernel<float>), dim3(B), dim3(256), 0, (hipStream_t)stream, s, active, cur_slot, ncand, cand_slot, Cmax,
                       (const float*)pano, pano_bs, pano_rs, (const float*)fused, fused_rs);
  else
    hipLaunchKernelGGL((graph_embed_kernel<bf16_t>), dim3(B), dim3(256), 0, (hipStream_t)stream, s, active, cur_slot, ncand, cand_slot, Cmax,
                       (const bf16_t*)pano, pano_bs, pano_rs, (const bf16_t*)fused, fused_rs);
  DSRG_LAUNCH_CHECK("dsrg_graph_update_embeds");
  return DSRG_OK;
}

extern "C" int dsrg_graph_nav_inputs(const dsrg_graph_state_t* st, const dsrg_graph_nav_out_t* out, const int32_t* cur_slot,
                                     const double* heading, const double* elevation, const int32_t* ncand, const int32_t* cand_slot,
                                     int Cmax, const void* mem, int mem_rs, int B, int dtype, void* stream) {
  DSRG_CHECK(st && out && cur_slot && heading && elevation && ncand && cand_slot, "dsrg_graph_nav_inputs: null pointer");
  DSRG_CHECK(dtype == DSRG_F32 || dtype == DSRG_BF16, "dsrg_graph_nav_inputs: bad dtype");
  DSRG_CHECK(st->N >= 1 && st->N <= MAXN && out->G >= 2 && out->G <= MAXN + 2 && out->Vp >= 2, "dsrg_graph_nav_inputs: bad sizes (N <= %d)", MAXN);
  DSRG_CHECK(out->gmap_img && out->gmap_step_ids && out->gmap_pos_fts && out->gmap_pair_dists && out->gmap_masks && out->gmap_visited &&
                 out->gmap_slots && out->gmap_src && out->bw_mask && out->vp_pos_fts && out->no_vp_left && out->gmap_len,
             "dsrg_graph_nav_inputs: null output");
  DSRG_CHECK(!mem || (mem_rs >= st->H && mem_rs % 4 == 0), "dsrg_graph_nav_inputs: bad mem stride");
  if (B == 0) return DSRG_OK;
  GraphState s{st->pos, st->dist, st->point, st->visited, st->step_id, st->emb_sum, st->emb_cnt, st->n_nodes, st->N, st->H};
  NavOut o{out->gmap_img, out->gmap_step_ids, out->gmap_pos_fts, out->gmap_pair_dists, out->gmap_masks, out->gmap_visited, out->gmap_slots,
           out->gmap_src, out->bw_mask, out->vp_pos_fts, out->no_vp_left, out->gmap_len, out->G, out->Vp};
  if (dtype == DSRG_F32)
    hipLaunchKernelGGL((graph_nav_inputs_kernel<float>), dim3(B), dim3(256), 0, (hipStream_t)stream, s, o, cur_slot, heading, elevation, ncand,
                       cand_slot, Cmax, (const float*)mem, mem_rs);
  else
    hipLaunchKernelGGL((graph_nav_inputs_kernel<bf16_t>), dim3(B), dim3(256), 0, (hipStream_t)stream, s, o, cur_slot, heading, elevation, ncand,
                       cand_slot, Cmax, (const bf16_t*)mem, mem_rs);
  DSRG_LAUNCH_CHECK("dsrg_graph_nav_inputs");
  return DSRG_OK;
}
