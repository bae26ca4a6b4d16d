/*
 * pt_amd.h -- C ABI of the MI355X-native path-tracing hot path (libptamd.so).
 *
 * This is the drop-in boundary for the per-GPU engine of 3DevApps/multi-gpu-path-tracer:
 * each entry point replaces one piece of the reference's C++ `DevicePathTracer` /
 * `Framebuffer` / `SceneLoader` / `TaskGenerator` surface (citations are file:line under
 * the reference's src/).  Plain pointers and sizes only: no HIP, torch or C++ types.
 *
 * Threading (mirrors DevicePathTracer.h:194-226 as used by StreamThread.h:64-104):
 * calls on one pt_device are externally serialised, EXCEPT pt_render_task_async /
 * pt_stream_sync, which may run concurrently from different host threads on different
 * streams of the same device.
 *
 * Errors: every call returns a pt_status (0 = PT_OK); pt_last_error_string() gives the
 * thread-local message.  Nothing throws across this boundary.  (The reference prints
 * "CUDA ERROR = <code> at <file>:<line>" and exit(99)s, cuda_utils.h:6-16; the C++ wrapper
 * in csrc/host/DevicePathTracer.h reproduces that on top of these codes.)
 */
#ifndef PT_AMD_H
#define PT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_API __attribute__((visibility("default")))

typedef int pt_status;
enum {
    PT_OK = 0,
    PT_ERR_INVALID = 1,      /* bad argument / out-of-range index                              */
    PT_ERR_HIP = 2,          /* a HIP runtime call failed                                      */
    PT_ERR_NO_LIGHTS = 3,    /* no emissive face: reference dereferences list[0] of an empty
                                light list (hitable_list.h:23-26) -> undefined behaviour        */
    PT_ERR_STACK = 4,        /* BVH needs > 64 traversal-stack entries (bvh.h:179 overflow, UB) */
    PT_ERR_IO = 5,           /* file missing / malformed                                       */
    PT_ERR_UNSUPPORTED = 6,  /* valid input this build does not handle                         */
    PT_ERR_NO_SCENE = 7,     /* render before pt_upload_scene / pt_bind_framebuffer            */
    PT_ERR_NO_DEVICE = 8,    /* no usable gfx950 device: the HIP path is mandatory, there is no
                                CPU fallback                                                   */
    PT_ERR_NOMEM = 9,        /* a host allocation failed (std::bad_alloc / length_error)        */
    PT_ERR_INTERNAL = 10     /* any other C++ exception, caught at the boundary: nothing is ever
                                thrown across this C ABI                                        */
};

/* ------------------------------------------------------------------ host data model -- */
/* The PACKED form of HostScene.h:28-31 Vertex: 20 bytes.  NOT layout-identical to the reference's struct: its
 * `float2 texCoords` is 8-byte aligned (CUDA's float2 is __builtin_align__(8), HIP's is alignas(8)), so the reference's
 * Vertex is 24 bytes with texCoords at offset 16, and its Triangle (HostScene.h:41-47) is 80 bytes.  Never cast a
 * `std::vector<Triangle>::data()` of the reference to `const pt_triangle *`: hand it over with pt_triangle_layout
 * (pt_upload_scene_strided / pt_scene_create_strided below), which reads each field at the caller's own offset.    */
typedef struct { float position[3]; float texCoords[2]; } pt_vertex;
/* The PACKED form of HostScene.h:41-47 Triangle: 68 bytes. */
typedef struct { pt_vertex v0, v1, v2; int textureIdx; int materialIdx; } pt_triangle;
/* Where the fields of the CALLER'S triangle record are: record stride and the byte offset of every field in it
 * (three floats per position, two per texCoords, one int each for the indices).  With the reference's headers:
 *   { sizeof(Triangle),
 *     { offsetof(Triangle, v0.position),  offsetof(Triangle, v1.position),  offsetof(Triangle, v2.position)  },
 *     { offsetof(Triangle, v0.texCoords), offsetof(Triangle, v1.texCoords), offsetof(Triangle, v2.texCoords) },
 *     offsetof(Triangle, textureIdx), offsetof(Triangle, materialIdx) }
 * which evaluates to PT_TRIANGLE_LAYOUT_REFERENCE under nvcc and hipcc alike.                                        */
typedef struct {
    size_t stride;
    size_t position[3];
    size_t texCoords[3];
    size_t textureIdx, materialIdx;
} pt_triangle_layout;
#define PT_TRIANGLE_LAYOUT_REFERENCE { 80, { 0, 24, 48 }, { 16, 40, 64 }, 72, 76 }
#define PT_TRIANGLE_LAYOUT_PACKED    { 68, { 0, 20, 40 }, { 12, 32, 52 }, 60, 64 }
/* HostScene.h:33-39 HostMaterial; a texture index of -1 is an empty std::optional<int>. */
typedef struct {
    float baseColor[3];       int baseColorTextureIdx;
    float emissiveFactor[3];  int emissiveTextureIdx;
} pt_material;
/* HostScene.h:49-53 HostTexture: width*height float3 texels with channel values 0..255. */
typedef struct { int width, height; const float *data; } pt_texture;
/* CameraConfig.h:5-17 (pitch / yaw are UI state, not read on this path). */
typedef struct { float lookFrom[3]; float front[3]; float vfov, hfov; } pt_camera;
/* DevicePathTracer.h:19-25 RenderTask (the `time` field stays on the C++ wrapper). */
typedef struct { int width, height, offset_x, offset_y; } pt_task;

/* Exact per-launch work counters: the kernel's own counts. */
typedef struct {
    uint64_t paths;        /* camera rays started                                   */
    uint64_t segments;     /* BVH::hit calls from camera::ray_color (camera.h:55)   */
    uint64_t box_tests;    /* aabb::hit evaluations (aabb.h:38)                     */
    uint64_t tri_tests;    /* triangle::hit evaluations inside BVH::hit             */
    uint64_t light_tests;  /* triangle::hit evaluations inside triangle::pdf_value  */
} pt_counters;

typedef struct {
    uint32_t nodes, leaves, max_leaf, depth, max_stack, lights, triangles;
} pt_bvh_info;

typedef struct pt_device pt_device;          /* one per GPU = one DevicePathTracer        */
typedef struct pt_host_scene pt_host_scene;  /* HostScene owned by the library            */
typedef void *pt_stream;                     /* hipStream_t; NULL = the null stream       */

/* ------------------------------------------------------------------------- errors -- */
PT_API const char *pt_last_error_string(void);
PT_API const char *pt_version(void);
/* Hardware queues the HIP runtime maps this process's streams onto = how many launches on different streams can overlap
 * (frames in flight; RenderManager.h:410-431 has no counterpart: the reference's renderFrame ends on a barrier).  The
 * library sets GPU_MAX_HW_QUEUES=8 when it is loaded unless the host has set it, and returns what the variable said then; a
 * process that initialised the HIP runtime before loading the library must export the variable itself. */
PT_API int pt_hw_queue_count(void);

/* --------------------------------------------------------- per-GPU engine (device) -- */
/* DevicePathTracer ctor, DevicePathTracer.h:169-192 (cudaSetDevice part). */
PT_API pt_status pt_device_create(int device_index, pt_device **out);
/* ~DevicePathTracer, DevicePathTracer.h:372-377 (frees everything, not only the BVH). */
PT_API pt_status pt_device_destroy(pt_device *dev);
/* reloadWorld(), DevicePathTracer.h:312-340: copies the scene (no host pointer is kept),
 * builds the reference-topology BVH and the light list.                                 */
PT_API pt_status pt_upload_scene(pt_device *dev,
                                 const pt_triangle *triangles, size_t n_triangles,
                                 const pt_material *materials, size_t n_materials,
                                 const pt_texture *textures, size_t n_textures);
PT_API pt_status pt_upload_host_scene(pt_device *dev, const pt_host_scene *scene);
/* reloadWorld() for a caller that keeps the reference's own `std::vector<Triangle>` (HostScene.h:41-47, 80-byte records
 * with 8-byte-aligned float2 members): `triangles` points at n_triangles records laid out as `layout` says.  Fields
 * are copied one by one into the packed form; padding bytes are never read.                                         */
PT_API pt_status pt_upload_scene_strided(pt_device *dev, const void *triangles, size_t n_triangles,
                                         const pt_triangle_layout *layout,
                                         const pt_material *materials, size_t n_materials,
                                         const pt_texture *textures, size_t n_textures);
/* setFramebuffer(), DevicePathTracer.h:342-358: rgb = 3*w*h bytes, yuv = w*h + 2*(w*h/4)
 * bytes (Framebuffer.h:27-35), both device-accessible.  No RNG table is allocated: the
 * per-pixel XORWOW state of render_init (DevicePathTracer.h:46-55) is re-derived in-kernel.
 * Cheap (no synchronisation): a launch renders into the buffers bound when it was enqueued, so
 * rebinding between launches is how several frames are kept in flight.                       */
PT_API pt_status pt_bind_framebuffer(pt_device *dev, int width, int height,
                                     uint8_t *rgb_device, uint8_t *yuv_device);
/* setThreadBlockSize(dim3), DevicePathTracer.h:368: a hint (pixel-tile shape per wave). */
PT_API pt_status pt_set_thread_block_size(pt_device *dev, unsigned x, unsigned y);
/* Acceleration structure built by the NEXT pt_upload_scene: 0 = the reference's topology
 * (bvh.h:20-176; results identical to the reference's traversal), 1 = opt-in SAH tree with
 * distance pruning: same Moller-Trumbore arithmetic and tie-breaking, but a
 * different set of tested triangles, so ill-conditioned rays can resolve differently.     */
PT_API pt_status pt_set_bvh_mode(pt_device *dev, int mode);
/* EXTENSION, off by default: the reference's legacy material classes and its sphere.
 * kind 0 = UniversalMaterial (the live path; albedo / param ignored), 1 = lambertian(albedo)
 * material.h:110-127, 2 = metal(albedo, fuzz = param) :130-144, 3 = dielectric(ir = param) :146-178,
 * 4 = diffuse_light(color = albedo) :210-217.  pt_sphere = sphere(center, radius, material) sphere.h:8-50;
 * spheres are tested after the BVH like members of a hitable_list.  No integrator of the reference at
 * HEAD calls these classes: the loop that drives them here is the classic one their scatter()/emitted()
 * signatures were written for (end on emitters and absorbed rays, else throughput *= attenuation and
 * follow the scattered ray); pow(x,5) in the Schlick term is ((x*x)*(x*x))*x.  PARITY UNPINNED; kernel and
 * oracle agree bit for bit.  The records apply to the NEXT pt_upload_scene (one per material; pass 0
 * materials and 0 spheres to switch the extension off); they need the reference tree (bvh mode 0).
 * Only faces with an emissive UniversalMaterial are light-sampled (DevicePathTracer.h:304-307), and a
 * scene without such a face is accepted only when nothing in it uses kind 0.                       */
typedef struct pt_legacy_material { int kind; float albedo[3]; float param; } pt_legacy_material;
typedef struct pt_sphere { float center[3]; float radius; int materialIdx; } pt_sphere;
PT_API pt_status pt_set_legacy_scene(pt_device *dev, const pt_legacy_material *materials, size_t n_materials,
                                     const pt_sphere *spheres, size_t n_spheres);

/* Cooperative tiles, for reference trees with big leaves walked by the wave-uniform stack walk (the duck:
 * one leaf of 1603 triangles): the four waves of a workgroup render the SAME 8x8 tile, each testing a
 * quarter of every leaf's triangles, and merge their closest hits once per ray segment (min over t, ties
 * by the reference's visit order).  Image and counters are unchanged; a tile's sequential chain gets ~3.5x
 * shorter for more total work, which pays when the launch is small for the GPU (a rank's share of a
 * multi-GPU frame).  -1 (default): automatic (leaf >= 64 triangles and tiles * 4 <= 3 x the device's wave
 * slots); 0: never; 1: whenever that walk is in use.                                              */
PT_API pt_status pt_set_cooperative_mode(pt_device *dev, int mode);

/* Order in which a launch hands its 8x8-pixel tiles to waves.  0: raster order.  1 (default):
 * cost order -- every launch records what each tile cost (shader-clock ticks) and a device radix
 * sort makes that the next launch's order for the same task rectangle / band set, dearest tiles
 * first, so a launch no longer ends on a straggler that started last.  This is the reference's
 * frame-to-frame feedback (DSFL/DSDL adapt task rectangles from last frame's task times,
 * RenderManager.h:264-408) applied inside one GPU at tile granularity.  Pixels never depend on the
 * order (per-pixel seeds), so images are byte-identical in both modes.  Up to 16 task geometries
 * are tracked per device; launches of fewer than 256 tiles always use raster order.
 * A schedule belongs to (task geometry, bound framebuffer): launches that share one are ordered
 * behind each other, so a caller that keeps several FRAMES IN FLIGHT -- pt_bind_framebuffer to
 * another buffer, launch on another stream -- gets one schedule per buffer and launches that
 * overlap on the device (the younger frame's waves take the wave slots the older frame's last
 * tiles leave idle; scheduling.py RenderManager(frames_in_flight=...)).                      */
PT_API pt_status pt_set_tile_schedule(pt_device *dev, int mode);

/* Traversal schedule of the render kernel (same results and test counts, different order of
 * work across the lanes of a wave): 0 = auto, 1 = per-lane two-phase walk (any
 * scene), 2 = wave-uniform walk with a 64-bit path mask (trees of <= 64 inner nodes),
 * 3 = wave-uniform walk with a per-lane LDS stack (<= 4096 inner nodes), 4 = the mask walk with the
 * triangle tests of sparsely hit leaves dealt evenly over the wave's 64 lanes through LDS (<= 64 inner
 * nodes and <= 1024 triangles), 5 = the stack walk with the same dealing (<= 4096 inner nodes and
 * <= 65536 triangles).                                                                     */
PT_API pt_status pt_set_traversal_mode(pt_device *dev, int mode);
/* renderTaskAsync(RenderTask&, cudaStream_t), DevicePathTracer.h:194-214: no-op when
 * task->width == 0; asynchronous on `stream`; camera by value as at :210.               */
PT_API pt_status pt_render_task_async(pt_device *dev, const pt_task *task,
                                      const pt_camera *camera, unsigned samples_per_pixel,
                                      unsigned recursion_depth, pt_stream stream);
/* Multi-GPU form of the above: ONE launch renders every row band owned by
 * `rank` (bands of band_rows framebuffer rows, band b belongs to rank b % n_ranks -- the tasks
 * pt_tasks_row_bands enumerates) into the bound full-frame buffers.                       */
PT_API pt_status pt_render_bands_async(pt_device *dev, int band_rows, int n_ranks, int rank,
                                       const pt_camera *camera, unsigned samples_per_pixel,
                                       unsigned recursion_depth, pt_stream stream);
/* The same with an explicit band list, for dealers that do not deal b % n_ranks (the cost-fed longest-processing-time
 * dealer of the multi-rank RenderManager, the analogue of the reference's frame-to-frame DSFL feedback,
 * RenderManager.h:334-408): ONE launch renders the listed bands (indices into the frame's bands of band_rows rows,
 * 0 = top; each at most once; the frame's partial last band, if listed, last).  `bands` is read before the call returns. */
PT_API pt_status pt_render_band_list_async(pt_device *dev, int band_rows, const int *bands, int n_bands,
                                           const pt_camera *camera, unsigned samples_per_pixel,
                                           unsigned recursion_depth, pt_stream stream);
/* What the bands of this device's latest row-band launch (either form) cost, in launch order: the sum over each band's
 * 8x8 tiles of the trip counts the render kernel recorded for the cost-ordered tile schedule -- deterministic, the same
 * whichever GPU renders the band.  Waits for that launch.  *n_bands = 0 when nothing was recorded (tile schedule off or
 * a launch of fewer than 256 tiles).                                                                        */
PT_API pt_status pt_band_costs_read(pt_device *dev, uint64_t *costs, int max_bands, int *n_bands);
/* ... of the latest row-band launch enqueued on `stream`: a host that keeps several frames in flight, each on its own
 * stream, reads the costs of the frame it has just completed (scheduling.py's cost-fed dealers under frames in flight). */
PT_API pt_status pt_band_costs_read_on(pt_device *dev, pt_stream stream, uint64_t *costs, int max_bands, int *n_bands);
/* synchronizeStream(cudaStream_t), DevicePathTracer.h:222-226. */
PT_API pt_status pt_stream_sync(pt_device *dev, pt_stream stream);
/* waitForRenderTask(), DevicePathTracer.h:216-220. */
PT_API pt_status pt_device_sync(pt_device *dev);
/* StreamThread.h:19-36: a worker's stream. */
PT_API pt_status pt_stream_create(pt_device *dev, pt_stream *out);
PT_API pt_status pt_stream_destroy(pt_device *dev, pt_stream stream);

/* Work counters (new; the metric's numerator).  Read after a sync. */
PT_API pt_status pt_counters_read(pt_device *dev, pt_counters *out);
PT_API pt_status pt_counters_reset(pt_device *dev);
/* All 8 raw counter words: [0..4] as pt_counters; [5] = blocks of 8 stack entries the persistent SAH kernel moved from
 * LDS to its overflow area in device memory (rare; tests use it to prove that path ran); [5..7] in the diagnostic
 * PT_STAMPS build (tools/stamps.sh): longest wave lifetime, summed traversal cycles, summed wave lifetimes, in
 * shader-clock cycles.                                                                    */
PT_API pt_status pt_debug_counters_read(pt_device *dev, uint64_t out8[8]);
/* How many of its dearest tiles the NEXT launch of the latest launch's rectangle hands to several waves each (SAH
 * kernel; 0 when the cost-ordered tile schedule is off, the launch was tiny or the walk is another).
 * Waits for the stream of that launch.                                                    */
PT_API pt_status pt_split_tiles_read(pt_device *dev, uint32_t *n_split);
/* Accumulated device time (ms) of the render kernel launches since the last reset,
 * measured with HIP events on the launch stream; n_launches may be NULL.                */
PT_API pt_status pt_kernel_time_read(pt_device *dev, double *ms_total, uint64_t *n_launches);
PT_API pt_status pt_kernel_time_reset(pt_device *dev);
/* Name of the gfx950 kernel the latest render launch of this device ran (the symbol a rocprofv3 kernel trace
 * shows), e.g. "pt::dev::render_kernel<false, true, 1>"; empty before the first launch.          */
PT_API pt_status pt_kernel_name_read(pt_device *dev, char *out, size_t cap);

/* BVH introspection (known-answer tests against the oracle).  bounds: 6 floats per node
 * {x.min,x.max,y.min,y.max,z.min,z.max}; links: 4 u32 per node {left,right,firstTri,triCount}
 * in the reference's node numbering (bvh.h:11-16, :78-90).                              */
PT_API pt_status pt_bvh_info_read(pt_device *dev, pt_bvh_info *out);
PT_API pt_status pt_bvh_copy(pt_device *dev, float *bounds6, uint32_t *links4, int *tri_indices,
                             int *light_faces);

/* Host-only build of the same BVH (create_world, DevicePathTracer.h:134-146 -> bvh.h:20-176)
 * without a device: bounds6 / links4 hold up to 2n-1 nodes, tri_indices n entries; any of the
 * three may be NULL.                                                                      */
PT_API pt_status pt_host_bvh_build(const pt_triangle *triangles, size_t n_triangles,
                                   pt_bvh_info *info, float *bounds6, uint32_t *links4,
                                   int *tri_indices);

/* Host-only build of the opt-in SAH tree (pt_set_bvh_mode(dev, 1)): pairs16 receives up to
 * n_triangles NodePair records of 16 dwords {L box: xmin xmax ymin ymax zmin zmax, R box: 6 floats,
 * llink lcnt rlink rcnt}; order[slot] = face stored at leaf-order slot `slot`.  info->nodes is the
 * number of pairs.  Either array may be NULL.                                             */
PT_API pt_status pt_host_sah_build(const pt_triangle *triangles, size_t n_triangles, pt_bvh_info *info,
                                   float *pairs16, int *order);

/* Framebuffer.h:27-35 replacement for cudaMallocManaged: device-local buffers + explicit
 * download (the reference relies on UVM page migration as its implicit gather).         */
PT_API pt_status pt_fb_alloc(pt_device *dev, int width, int height, uint8_t **rgb_device,
                             uint8_t **yuv_device);
PT_API pt_status pt_fb_free(pt_device *dev, uint8_t *rgb_device, uint8_t *yuv_device);
PT_API pt_status pt_fb_download(pt_device *dev, const uint8_t *src_device, uint8_t *dst_host,
                                size_t bytes, pt_stream stream);

/* The explicit gather that replaces UVM page migration in single-process use: copies the
 * pixels of `task` (RGB rows, Y rows and the U/V samples it owns) from the device framebuffer
 * bound with pt_bind_framebuffer into full-frame host buffers of the same layout, then waits. */
PT_API pt_status pt_fb_download_task(pt_device *dev, const pt_task *task, uint8_t *rgb_host,
                                     uint8_t *yuv_host, pt_stream stream);

/* ---- native gather (north_star: "final RCCL gather of tile spans over xGMI", from C++ without Python) ---------------
 * Exchange layout of a list of task rectangles of one framebuffer: plane-major over the list -- the RGB rows of every
 * task, then their Y rows, then the U and the V samples they own (even framebuffer rows / columns, Framebuffer.h:27-35's
 * planes).  For full-width row bands it is the byte string of the Python path's scheduling.pack_bands.               */
PT_API pt_status pt_fb_packed_size(int width, int height, const pt_task *tasks, int n_tasks, size_t *bytes);
/* Pack the rectangles out of / scatter them into the device's bound framebuffer, on `stream`; packed_device is DEVICE
 * memory of pt_fb_packed_size bytes.  Asynchronous with respect to the kernels (the task list is read before return). */
PT_API pt_status pt_fb_pack_tasks(pt_device *dev, const pt_task *tasks, int n_tasks, uint8_t *packed_device, pt_stream stream);
PT_API pt_status pt_fb_unpack_tasks(pt_device *dev, const pt_task *tasks, int n_tasks, const uint8_t *packed_device, pt_stream stream);
/* One process driving N GPUs (RenderManager.h:76-112: a DevicePathTracer per GPU): the replacement for the reference's
 * implicit gather through CUDA managed memory (Framebuffer.h:27-35).  pt_gather_create makes one RCCL communicator per
 * device (ncclCommInitAll; librccl is loaded on demand, a single device needs none; see pt_gather_transport for the
 * fallback).  pt_gather_frame, called once per
 * frame after every render stream has been synchronised: device task_device[i] (an index into `devices`) rendered
 * tasks[i]; every device but the first packs its rectangles, ONE ncclGroupStart / ncclSend + ncclRecv per peer /
 * ncclGroupEnd moves the strings to the first device over each peer's own xGMI link, the first device scatters them
 * into its framebuffer and the whole frame reaches rgb_host / yuv_host with one pair of copies.  All devices must have
 * framebuffers of the same size bound.                                                                               */
typedef struct pt_gather pt_gather;
PT_API pt_status pt_gather_create(pt_device *const *devices, int n_devices, pt_gather **out);
PT_API pt_status pt_gather_destroy(pt_gather *gather);
/* How the strings travel: "rccl", or "peer-copy (<why>)" -- the fallback when librccl cannot be loaded, ncclCommInitAll
 * fails, PTAMD_GATHER=peer is set or one GPU is listed more than once: the same packed strings moved with one
 * hipMemcpyPeerAsync per peer (xGMI where peers have direct access) -- or "none" for a single device.              */
PT_API pt_status pt_gather_transport(const pt_gather *gather, char *out, size_t cap);
PT_API pt_status pt_gather_frame(pt_gather *gather, const pt_task *tasks, const int *task_device, int n_tasks,
                                 uint8_t *rgb_host, uint8_t *yuv_host);

/* Device arithmetic self-test (bit-exactness contract): evaluates on the GPU
 * op 0: sincos(x)->(out0,out1); 1: a/b; 2: sqrt(a); 3: (float)((double)a / pi);
 * 4: XORWOW uniform #b (0-based) of seed (uint32)a bits -> out0.                          */
PT_API pt_status pt_kat_arith(pt_device *dev, int op, const float *a, const float *b,
                              float *out0, float *out1, size_t n);
/* Exhaustive proof obligations of the kernels' short division forms (pt_math.h): runs the bit
 * patterns first_bits .. first_bits+count-1 (count <= 2^32) through the short form and through the
 * IEEE expression it replaces (op 0: 1.0f/x; op 1: (float)((double)x / pi); op 2: 1.0f/x for
 * |x| > 1e-8 only, the Moller-Trumbore form; op 3: the unguarded core, which must fail; op 4: sqrtf(x)) on the
 * GPU; *n_bad =
 * number of inputs whose results differ in any bit (NaN == NaN), the first max_examples of them
 * (unordered) are written to examples.                                                        */
PT_API pt_status pt_kat_exhaustive(pt_device *dev, int op, uint32_t first_bits, uint64_t count,
                                   uint64_t *n_bad, uint32_t *examples, int max_examples);
/* Closest hit of n rays through the device traversal: o/d are 3 floats per ray; outputs
 * t (or -1 on miss) and the face index in the uploaded triangle order (-1 on miss).     */
PT_API pt_status pt_kat_closest_hit(pt_device *dev, const float *origins, const float *dirs,
                                    float *t_out, int *face_out, size_t n);
/* Diagnostic companion of pt_kat_closest_hit: the same walk, and per ray how many box and triangle tests it took
 * (either output may be NULL).  Wave-uniform walks (big reference trees) report wave totals in every lane.  */
PT_API pt_status pt_kat_walk_cost(pt_device *dev, const float *origins, const float *dirs,
                                  uint32_t *box_tests_out, uint32_t *tri_tests_out, size_t n);

/* --------------------------------------------------------------- scene I/O (host) -- */
/* SceneLoader::load(std::string&), HostScene.cpp:98-139, for glTF 2.0 input: a .glb container or .gltf
 * JSON text (buffers / images as files next to it, base64 data: URIs or bufferViews).  Node transforms
 * are baked like aiProcess_PreTransformVertices; textures are PNG or JPEG, decoded to exactly what the
 * reference's vendored stb_image returns (embedded: native channels + the reference's 3-bytes-per-pixel
 * copy, HostScene.cpp:19-26,36-46; image files: req_comp = 3, :28-30).                              */
PT_API pt_status pt_scene_load_glb(const char *path, pt_host_scene **out);
/* Wavefront OBJ (+ name-prefix material rule of obj_loader.h:65-96 / README.md:60-76). */
PT_API pt_status pt_scene_load_obj(const char *path, pt_host_scene **out);
PT_API pt_status pt_scene_create(const pt_triangle *triangles, size_t n_triangles,
                                 const pt_material *materials, size_t n_materials,
                                 const pt_texture *textures, size_t n_textures,
                                 pt_host_scene **out);
/* pt_scene_create for triangle records in the caller's own layout (see pt_triangle_layout). */
PT_API pt_status pt_scene_create_strided(const void *triangles, size_t n_triangles, const pt_triangle_layout *layout,
                                         const pt_material *materials, size_t n_materials,
                                         const pt_texture *textures, size_t n_textures,
                                         pt_host_scene **out);
/* The conversion both of the above use, on its own: n records at `triangles` in `layout` -> packed pt_triangle[n]. */
PT_API pt_status pt_triangles_repack(const void *triangles, size_t n_triangles, const pt_triangle_layout *layout,
                                     pt_triangle *out);
PT_API void      pt_scene_free(pt_host_scene *scene);
PT_API size_t    pt_scene_triangle_count(const pt_host_scene *scene);
PT_API size_t    pt_scene_material_count(const pt_host_scene *scene);
PT_API size_t    pt_scene_texture_count(const pt_host_scene *scene);
PT_API const pt_triangle *pt_scene_triangles(const pt_host_scene *scene);
PT_API const pt_material *pt_scene_materials(const pt_host_scene *scene);
PT_API const pt_texture  *pt_scene_textures(const pt_host_scene *scene);
PT_API const char *pt_scene_material_name(const pt_host_scene *scene, size_t index);
/* EXTENSION: the legacy class obj_loader.h:65-96 picks for each material by name prefix -- lambertian*
 * (albedo = Ka), metal* (albedo = Ka, fuzz = Ns), dielectric* (ir = Ni), diffuse_light* (color = Kd) -- as
 * records for pt_set_legacy_scene; kind 0 for every other name and for glTF input.  Returns the number of
 * materials and copies at most `cap` records.                                                       */
PT_API size_t    pt_scene_legacy_materials(const pt_host_scene *scene, pt_legacy_material *out, size_t cap);
/* Benchmark convention for models/cornell_box.glb (BASELINE.md note 1):
 * (i) zero-emissive materials named diffuse_light* get emissiveFactor (1,1,1);
 * (ii) exact axis swap (x,y,z) -> (x,z,-y) on every vertex.                             */
PT_API pt_status pt_scene_apply_cornell_convention(pt_host_scene *scene);
/* Deterministic synthetic terrain of 2*grid*grid triangles + one emissive quad
 * (BASELINE.json config 4 uses grid = 2048 -> 8,388,608 + 2 triangles).                 */
PT_API pt_status pt_scene_make_terrain(unsigned grid, pt_host_scene **out);
/* Binary PPM (P6) sink for the .glb -> PPM pipeline (README.md:54-58 names build/out.ppm). */
PT_API pt_status pt_write_ppm(const char *path, int width, int height, const uint8_t *rgb);

/* ------------------------------------------------------------ scheduling (host) ---- */
/* TaskGenerator::generateEqualTasks(taskCount, taskLayout, w, h), TaskGenerator.h:58-80,
 * with RenderManager::getTaskLayout(maxTasksInRow), RenderManager.h:42-59 (FSFL).       */
PT_API pt_status pt_tasks_equal_layout(int task_count, int max_tasks_in_row, int width,
                                       int height, pt_task *out_tasks);
/* TaskGenerator::generateEqualTasks(task_count, w, h), TaskGenerator.h:46-55 (one row). */
PT_API pt_status pt_tasks_equal_row(int task_count, int width, int height, pt_task *out_tasks);
/* Row-band partition used by the multi-GPU path: the frame is cut into
 * full-width bands of `band_rows` fb rows (multiple of 2); band b belongs to rank b % n.
 * Returns the bands of `rank` as tasks (offset_y counted from the image bottom, as
 * RenderTask does) and their count.                                                     */
PT_API pt_status pt_tasks_row_bands(int width, int height, int band_rows, int n_ranks, int rank,
                                    pt_task *out_tasks, int max_tasks, int *n_out);

#ifdef __cplusplus
}
#endif
#endif /* PT_AMD_H */
