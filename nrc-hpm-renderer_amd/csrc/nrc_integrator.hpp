// nrc_integrator.hpp -- volumetric path integrator kernels (host launch interface).
// HIP restatement of data/shader/nrc/{clear,gen_rays,prep_infer_rays,prep_train_rays,render}.comp,
// data/shader/mc/render.comp and data/shader/include/{random,volume,dir_gen,path_trace}.glsl.
#pragma once
#include "nrc_common.hpp"

namespace nrc {

// device-resident scene constants (specialization constants + UBOs of the reference: nrc-constants.glsl:18-26,
// nrc-descriptors.glsl:13-38)
struct DevScene {
    const uint8_t* density;       // R8, index i + nx*(j + ny*k)
    uint32_t nx, ny, nz;
    float fnx, fny, fnz;
    float size[3], half_size[3], inv_size[3];
    float len2size;               // length(2*skySize)
    float density_factor, inv_max_density, g;
    float dir_light_dir[3], dir_light_strength;
    float point_light_pos[3], point_light_strength, point_light_color[3];
    const float* env;             // RGBA32F
    uint32_t env_w, env_h;
    float env_strength;
    // exact occupancy of the volume, one bit per cell of (1 << occ_shift)^3 voxels (set: the cell holds a non-zero voxel), cell index
    // (cz * occ_gy + cy) * occ_gx + cx; at most kOccMaxWords words, copied into LDS by every kernel that samples the volume.
    // nullptr: no occupancy information (every look-up goes to memory)
    const uint32_t* occ_bits;
    uint32_t occ_shift, occ_gx, occ_gy, occ_words;
};
constexpr uint32_t kOccMaxWords = 2048;      // 65 536 cells = 8 KB of LDS

struct DevCamera {
    float m[16];                  // invProjView, column-major
    float pos[3];
};

struct DevFrame {
    float random[4];              // UniformData.random
    uint32_t w, h;                // local image: w columns x h rows
    uint32_t x_offset, x_stride;  // global x = ((x_offset + (lx >> x_block_log2) * x_stride) << x_block_log2) + (lx & block - 1)
    uint32_t x_block_log2;        // log2 of the strip width (nrc_tile::x_block)
    float inv_gw, inv_gh;         // 1/global width, 1/global height (ONE_OVER_RENDER_WIDTH/HEIGHT)
    // one bit per 8x8 pixel tile (index ty * ceil(w/8) + tx), set when a camera ray of the tile can meet a non-empty voxel; the
    // word behind the last tile word is non-zero when the mask must be ignored.  nullptr: no mask (every tile is traced).
    const uint32_t* tile_mask;
    // the mask may only be applied to a pixel whose delta walk through empty space provably leaves the volume before DeltaTrack's
    // cap of 128 collisions -- a property of the pixel's RNG state (2^23 of them) and of the scene's largest optical depth.  The
    // states that can reach the cap ("capped") are handed over as a list when there are at most kFlightListMax of them (one state,
    // the chain's fixed point 0, for the bench scene), otherwise as a bit per state (flight_bits, 1 MB, bit set = capped).
    // flight_mode: 0 the mask is not applied, 1 list, 2 bits.
    uint32_t flight_mode, flight_n;
    uint32_t flight_list[8];
    const uint32_t* flight_bits;
    // launch order of the camera kernels' 8x8 tiles: launch slot (workgroup * 4 + wave) -> default slot (the centre-out order of
    // pixel_of_wave_tile), costliest first (k_tile_order); nullptr: default order.  tile_cost: cycles each default slot's wave
    // took in this launch (written by k_gen_rays when non-null)
    const uint32_t* tile_order;
    uint32_t* tile_cost;
    // 0: tile_cost[slot] = this launch's cycles; k > 0: max(this launch's cycles, old - (old >> k)) -- a decaying maximum over the
    // sampled launches: the costliest-first order is hurt by tiles it under-estimates (a long tile started late ends the launch),
    // not by tiles it over-estimates
    uint32_t tile_cost_keep;
    // tiles k_gen_rays starts FIRST, whatever the order says: {kHotTilesMax entries (ty << 16 | tx), then their count}, written by
    // the previous frame's k_gen_rays (hot_next) or by k_hot_tiles for this frame's random numbers; nullptr: none.  A pixel in a capped RNG state (see flight_mode) in a tile the
    // mask rejects is one lane that walks for ~120 us; its wave would otherwise start among the empty tiles at the end of the
    // launch and end it that much later (one frame in four on the bench view).  Scheduling only: every tile is traced exactly once
    // with or without the list.
    const uint32_t* hot_tiles;
    // k_gen_rays builds the NEXT frame's list itself: every wave tests its tile's pixels against the next frame's random numbers
    // (random_next; one more hash per pixel, the pixel's own part of the seed is shared) and appends to hot_next; it also zeroes the
    // count cell of the list after that one (hot_reset).  Three list buffers rotate; no extra launch, no event.  nullptr: off.
    uint32_t* hot_next;
    uint32_t* hot_reset;
    float random_next[4];
    // the frame's LIVE queries (pixels that scattered: 22 % on the bench view): k_gen_rays appends their query indices to live_list
    // and counts them in *live_count (zeroed by the caller in front of the launch); nullptr: no list.  The order of the
    // list is whatever the waves' atomics made it; its only reader (the HashGrid encoder, Mlp::launch_features) writes by query index.
    uint32_t* live_list;
    uint32_t* live_count;
    // nrc_common.hpp's run-time priority switch as a kernel argument of the camera kernels (set by their launchers): their 32 400 waves
    // would each pay two dependent scalar loads for the device-side copy (0.7 % of a launch)
    uint32_t raise_priority;
    // set by the renderer, read by the launchers only: 1 = this renderer's camera kernels stay at the default wave priority although the
    // library raises (they then yield issue slots to the inference / training kernels beside them; Renderer::camera_priority_low_)
    uint32_t camera_priority_low;
    // the queries of pixels that did not scatter are not written (nobody reads them: the inference walks live_list and encodes inside its
    // kernel or from the list).  0: every slot of the query buffer is written, dead pixels with zeros (the reference's zero-filled buffer)
    uint32_t skip_dead_queries;
};
constexpr uint32_t kOrderSlotMask = 0x00ffffffu;      // (most tiles a launch order can address)
constexpr uint32_t kHotTilesMax = 8;      // = the waves of the two workgroups the launch gains in front

// forward camera transform for the tile mask: clip = m * (x, y, z, 1), column-major like DevCamera::m
struct DevProjView {
    float m[16];
};

struct TrainGrid {
    uint32_t tw, th, x_dist, y_dist, spp, ray_length, ring_size;
};

// origin / dir (the NRC vertex images) are written at the train grid's pixels only unless full_vertex_images
void launch_gen_rays(const DevScene& sc, const DevCamera& cam, const DevFrame& fr, uint32_t primary_ray_length,
                     float primary_ray_prob, float* primary, float* info, float* origin, float* dir, float* infer_in,
                     unsigned long long* fetch_counter, const TrainGrid& tg, bool full_vertex_images, hipStream_t s);

// marks the 8x8-pixel tiles whose camera rays can meet a non-empty voxel: `boxes` = n axis-aligned world-space boxes
// {lo.xyz, hi.xyz} that together cover every non-empty voxel with a margin of one voxel.  mask: ceil(tiles/32) + 1 words, zeroed.
// finds the pixels of the frame whose RNG state is in fr.flight_list (flight_mode 1) and appends their tiles to hot (count zeroed by the caller: hot[kHotTilesMax])
void launch_hot_tiles(const DevFrame& fr, uint32_t* hot, hipStream_t s);
// tile-major query order of the renderer's inference buffers (see query_index in nrc_integrator.hip) -> x * H + y
void launch_query_layout(const DevFrame& fr, uint32_t floats_per_query, const float* tiled, float* linear, hipStream_t s,
                         const float* info = nullptr);
uint32_t query_count(uint32_t w, uint32_t h);      // queries in tile-major order: whole 8x8 tiles
void launch_tile_mask(const float* boxes, uint32_t n_boxes, const DevProjView& pv, const DevFrame& fr, uint32_t* mask, hipStream_t s);
uint32_t tile_mask_words(uint32_t w, uint32_t h);
// table[m] = optical distance covered by the 128 free flights a delta walk draws from RNG state m when it rejects every collision
constexpr uint32_t kFlightStates = 1u << 23;
constexpr uint32_t kFlightListMax = 8;
void launch_flight_table(float* table, hipStream_t s);
// the states whose table entry does not exceed lambda: count_and_list[0] = their number, [1..8] the first eight (any order);
// bits (kFlightStates / 32 words): bit m set for every such state
void launch_flight_select(const float* table, float lambda, uint32_t* count_and_list, uint32_t* bits, hipStream_t s);
// launch slots of the camera kernels (rows padded to an odd number of workgroups) and the costliest-first order over them
uint32_t camera_slots(uint32_t w, uint32_t h);
void launch_tile_order(const uint32_t* cost, uint32_t n_slots, uint32_t* order, uint32_t w, hipStream_t s, uint32_t xcd_window = 0, uint32_t workgroups_in_front = 0);

void launch_mc_render(const DevScene& sc, const DevCamera& cam, const DevFrame& fr, uint32_t path_length,
                      float blend_factor, float* out_rgba, float* info, unsigned long long* fetch_counter, hipStream_t s);

// ring: {uint head, uint tail, RayInfo[ring_size]}; scratch: uint32[2*T + 4]
// start == nullptr: the whole of prep_train_rays.comp (scan, pop / trace / write, ring push).  start != nullptr (long train paths, the split
// frame graph): scan, the rays' start vertices -> start ([T][6] floats) + the training inputs, ring push; launch_train_trace does the rest
void launch_prep_train(const DevScene& sc, const DevFrame& fr, const TrainGrid& tg, const float* info,
                       const float* origin, const float* dir, uint32_t* ring, uint32_t* scratch, float* train_in,
                       float* train_target, hipStream_t s, float* start = nullptr);
void launch_train_trace(const DevScene& sc, const DevFrame& fr, const TrainGrid& tg, const float* start, float* train_target, hipStream_t s);
bool train_paths_are_long(const TrainGrid& tg);      // train ray length x spp >= 4 (quirk Q2 fixed, or a reference build with longer paths)

void integrator_set_wave_priority_raise(int on);      // nrc_common.hpp: NRC_RAISE_WAVE_PRIORITY's run-time switch, this file's kernels
void launch_composite(const DevFrame& fr, uint32_t show_nrc, float blend_factor, const float* primary,
                      const float* info, const float* infer_out, float* out_rgba, hipStream_t s, uint32_t* live_count_reset = nullptr);

// Reference::Result; d_scratch: double[8 + 2048]
void launch_compare(const float* ref_rgba, const float* own_rgba, uint32_t n_pixels, double* d_scratch, float* d_result5,
                    hipStream_t s);

// the metrics of a frame sharded over ranks: _1 leaves this rank's raw sums in d_scratch[0..3] (all-reduce them), _2 this rank's raw
// variance sum around the global mean in d_scratch[4] (all-reduce it), _3 writes the Result
void launch_compare_sharded_1(const float* ref_rgba, const float* own_rgba, uint32_t n_pixels, double* d_scratch, hipStream_t s);
void launch_compare_sharded_2(const float* ref_rgba, const float* own_rgba, uint32_t n_pixels, double* d_scratch, hipStream_t s);
void launch_compare_sharded_3(double* d_scratch, float* d_result5, hipStream_t s);
// a sharded frame's column strips: local [h][lw] -> [h][max_lw]; gathered [world][h][max_lw] -> [h][gw]
void launch_pad_columns(const float* local, uint32_t lw, uint32_t h, uint32_t max_lw, float* padded, hipStream_t s);
void launch_assemble_columns(const float* gathered, uint32_t world, uint32_t block_log2, uint32_t gw, uint32_t h, uint32_t max_lw, float* out,
                             hipStream_t s);
void launch_test_math(int fn, const float* a, const float* b, uint32_t n, float* out, float* out2, hipStream_t s);
void launch_test_rng(float u, float v, const float* frame_random4, uint32_t n, float* out, hipStream_t s);

}  // namespace nrc
