/* rt_hip.h -- C ABI of librt_hip.so, the MI355X (gfx950) render path for mkagie/ray_tracing scenes.
 *
 * The reference has no FFI of its own: its render path is the inline loop of src/main.rs:531-599,
 * which owns a tree of Box<dyn Hittable> and calls Ray::get_color (src/lib.rs:56-77) per sample.
 * Trait objects cannot cross to a GPU, so the boundary is cut in two places:
 *
 *   1. Scene construction ("rtb_*"): one entry point per constructor of the reference's prelude
 *      (src/lib.rs:23-35).  A Rust-side `flatten(&self, b: *mut RtBuilder) -> RtObj` method beside
 *      Hittable::try_hit (src/objects.rs:16-22) would call these while walking the tree.
 *   2. Rendering ("rt_*"): replaces the pixel/sample loop and everything below it
 *      (src/main.rs:567-599): inputs are the flattened world, the camera and the frame
 *      parameters; output is the per-pixel colour SUM that src/main.rs:585-589 computes.
 *
 * All functions return RT_OK (0) or a negative RtStatus; rt_last_error() gives the message of the
 * calling thread's last failure (the reference panics instead: src/main.rs:511-512, bvh.rs:140).
 * Nothing unwinds across this boundary.  Handles are opaque; inputs are copied; outputs are
 * caller-allocated.  Plain C, no HIP or torch types in any signature. */
#ifndef RT_HIP_H
#define RT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum RtStatus {
    RT_OK = 0,
    RT_ERR_INVALID = -1,     /* bad argument / handle */
    RT_ERR_HIP = -2,         /* a HIP runtime call failed (message has the HIP error string) */
    RT_ERR_NO_DEVICE = -3,   /* no usable GPU */
    RT_ERR_OOM = -4,
    RT_ERR_NO_BBOX = -5,     /* "No bounding box in bvh constructor", src/bvh.rs:140,152,201 */
    RT_ERR_UNSUPPORTED = -6, /* a tree the flat form cannot express */
    RT_ERR_CONFIG = -7,      /* "Could not open config file." / "Could not read values.", src/main.rs:511-512 */
    RT_ERR_NCCL = -8,        /* an RCCL call of rt_render_multi failed, or librccl could not be loaded */
    RT_ERR_INTERNAL = -9     /* an invariant of the library failed (a ray queue overflowed): the output is not to be used */
} RtStatus;

typedef struct RtBuilder RtBuilder; /* host scene graph under construction */
typedef struct RtScene RtScene;     /* flattened scene resident on one GPU */
typedef int32_t RtTex;              /* handles into a builder; < 0 = error */
typedef int32_t RtMat;
typedef int32_t RtObj;

const char* rt_last_error(void);
int rt_device_count(int* count);

/* ------------------------------------------------------------------ scene construction */
int rtb_create(RtBuilder** out);
void rtb_destroy(RtBuilder* b);
int rtb_object_count(const RtBuilder* b);   /* prims + media created so far (= next object id) */

/* textures -- src/textures.rs */
RtTex rtb_solid_color(RtBuilder* b, double r, double g, double bl);                 /* SolidColor::new  :24 */
RtTex rtb_checker(RtBuilder* b, RtTex even, RtTex odd);                             /* Checker::new     :40 */
/* Noise::new :80 -- the caller passes the Perlin tables it drew (:110-128): 256 xyz vectors and
 * perm_x|perm_y|perm_z (3 x 256 values in 0..255). */
RtTex rtb_noise(RtBuilder* b, double scale, const double* ranvec_768, const int32_t* perm_768);
/* materials -- src/materials.rs */
RtMat rtb_lambertian(RtBuilder* b, RtTex albedo);                                   /* Lambertian::from_texture :67 */
RtMat rtb_metal(RtBuilder* b, double r, double g, double bl, double fuzz);          /* Metal::new       :116 */
RtMat rtb_dielectric(RtBuilder* b, double ir);                                      /* Dielectric::new  :164 */
RtMat rtb_diffuse_light(RtBuilder* b, RtTex emit);                                  /* DiffuseLight::new :230 */
RtMat rtb_isotropic(RtBuilder* b, RtTex albedo);                                    /* Isotropic::new   :263 */
/* objects -- src/objects.rs, src/transrot.rs, src/mediums.rs, src/bvh.rs */
RtObj rtb_sphere(RtBuilder* b, const double center[3], double radius, RtMat m);     /* Sphere::new      :161 */
RtObj rtb_moving_sphere(RtBuilder* b, const double c0[3], const double c1[3], double t0, double t1, double radius,
                        RtMat m);                                                   /* MovingSphere::new :266 */
/* Rectangle::new :363 -- type 0 = Xy, 1 = Xz, 2 = Yz (:339-343) */
RtObj rtb_rectangle(RtBuilder* b, RtMat m, double x0, double x1, double y0, double y1, double k, int type);
RtObj rtb_box(RtBuilder* b, const double box_min[3], const double box_max[3], RtMat m);      /* BoxObj::new :442 */
RtObj rtb_list(RtBuilder* b, const RtObj* children, int n);                         /* HittableList     :25 */
RtObj rtb_bvh_node(RtBuilder* b, RtObj list, double time0, double time1);           /* BvhNode::new     bvh.rs:89 */
RtObj rtb_translate(RtBuilder* b, RtObj obj, const double offset[3]);               /* Translate::new   transrot.rs:15 */
RtObj rtb_rotate_y(RtBuilder* b, RtObj obj, double angle_deg);                      /* RotateY::new     transrot.rs:53 */
RtObj rtb_constant_medium(RtBuilder* b, RtObj boundary, double density, RtTex albedo);       /* ConstantMedium::new mediums.rs:23 */

/* ------------------------------------------------------------------ host side of main.rs
 * (what the Rust binary does before the render loop; provided in C++ because no Rust toolchain
 * exists where this library is built) */
typedef struct RtConfig {                       /* Config, src/main.rs:453-467 */
    int32_t image_width, image_height, samples_per_pixel, max_depth;
    double look_from[3], look_at[3], v_up[3];   /* CameraConfig, src/cameras.rs:10-18 */
    double vertical_fov_deg, aspect_ratio, aperture, focus_distance;
    char scene_type[32];                        /* enum Scene tag, src/main.rs:483-494 */
    int32_t n_list_objects;
    /* optional keys that the reference hard-codes (src/main.rs:534, 544-545): camera.shutter_open / shutter_close (0, 1) and
     * a top-level background {x, y, z} (black) */
    double shutter_open, shutter_close;
    float background[3];
} RtConfig;
/* Reads a config file (src/main.rs:510-516) and builds its scene (src/main.rs:520-529) into `b`.  A `List` scene accepts the
 * reference's items (spheres with Lambertian / Metal / Dielectric, src/objects.rs:102-107) and typed items for every other
 * constructor of the prelude: MovingSphere, Rectangle, Box, Translate, RotateY, ConstantMedium, nested List; DiffuseLight and
 * Isotropic materials; SolidColor / Checker / Noise textures (the schema is described in ray_tracing_amd/csrc/config.cpp).
 * `scene_seed` seeds the stream that replaces thread_rng in the canned scenes.  *world receives the
 * HittableList that main.rs:531 would wrap in BvhNode::new. */
int rt_config_load(const char* path, RtConfig* out);
int rt_config_build_scene(const char* path, uint64_t scene_seed, RtBuilder* b, RtObj* world);
/* One of Random | TwoSpheres | TwoPerlinSpheres | SimpleLight | Cornell | CornellSmoke | FinalScene (src/main.rs:483-494),
 * or "Synthetic:<n>" (not in the reference): n random spheres over a 2000 x 2000 field, the deep-BVH traversal stress scene. */
int rt_canned_scene(RtBuilder* b, const char* scene_type, uint64_t scene_seed, RtObj* world);

/* ------------------------------------------------------------------ flattened scene */
/* Flattens the tree under `world` (BVH build included, replacing BvhNode::new of src/main.rs:531)
 * and uploads it to GPU `device`. */
int rt_scene_create(const RtBuilder* b, RtObj world, int device, RtScene** out);
/* flags = RT_BVH_LINEAR: the top-level BVH (what BvhNode::new of src/main.rs:531 is in the reference, src/bvh.rs:89-156) is a
 * linear BVH built on the GPU -- Morton codes, radix sort, radix tree, bottom-up boxes -- instead of the host's binned-SAH
 * tree: milliseconds instead of seconds for a million primitives, at the price of a slower tree to traverse.  Closest hits
 * do not depend on the tree.  rt_flat_create_ex runs the same steps on the host. */
#define RT_BVH_LINEAR 1
int rt_scene_create_ex(const RtBuilder* b, RtObj world, int device, int flags, RtScene** out);
void rt_scene_destroy(RtScene* s);
typedef struct RtSceneInfo {
    int32_t n_nodes, n_spheres, n_moving_spheres, n_rects, n_instances, n_media, n_materials, n_textures, n_perlin;
    int32_t tlas_depth, blas_depth, max_stack, n_objects, device, n_boxes;  /* n_boxes: BoxObj kept as one 6-sided primitive */
    double world_box[6];
    uint64_t device_bytes;
    double bvh_build_seconds;   /* RT_BVH_LINEAR scenes: device time of the top-level build */
    double bvh_refit_seconds;   /* device time of the last rt_scene_refit */
} RtSceneInfo;
int rt_scene_info(const RtScene* s, RtSceneInfo* out);
/* Moving primitives.  The reference has no such path: a changed world is a new BvhNode::new (src/main.rs:531, src/bvh.rs:89-156).
 * Here spheres and moving spheres of a resident scene are rewritten in place -- ids are object ids in creation order
 * (rtb_object_count); xyzr = 4 doubles per sphere (centre, radius, src/objects.rs:161); a moving sphere takes 9 doubles
 * (centre0, centre1, time0, time1, radius, src/objects.rs:266) -- and rt_scene_refit recomputes every box of the scene's trees
 * bottom-up on the GPU, keeping their shape (boxes as src/objects.rs:241-244, 328-335, 413-429, 514-516 and
 * src/transrot.rs:34-38, 60-82 define them).  Closest hits after a refit equal those of a scene built from the moved
 * primitives; only the traversal cost depends on how far things moved.  A primitive that bounds a constant medium cannot be
 * moved (RT_ERR_UNSUPPORTED).  rt_flat_set_* / rt_flat_refit do the same on a host-only flatten. */
int rt_scene_set_spheres(RtScene* s, const int32_t* object_ids, int32_t n, const double* xyzr);
int rt_scene_set_moving_spheres(RtScene* s, const int32_t* object_ids, int32_t n, const double* c0_c1_t0_t1_r);
int rt_scene_refit(RtScene* s);

/* Host-only flatten (no GPU touched): for inspection and CPU-side checks.  The returned pointer is
 * an RtSceneView (ray_tracing_amd/csrc/rt_types.h) with HOST pointers, valid until rt_flat_destroy. */
typedef struct RtFlat RtFlat;
int rt_flat_create(const RtBuilder* b, RtObj world, RtFlat** out);
int rt_flat_create_ex(const RtBuilder* b, RtObj world, int flags, RtFlat** out);
const void* rt_flat_view(const RtFlat* f);
int rt_flat_info(const RtFlat* f, RtSceneInfo* out);
void rt_flat_destroy(RtFlat* f);
int rt_flat_set_spheres(RtFlat* f, const int32_t* object_ids, int32_t n, const double* xyzr);
int rt_flat_set_moving_spheres(RtFlat* f, const int32_t* object_ids, int32_t n, const double* c0_c1_t0_t1_r);
int rt_flat_refit(RtFlat* f);

/* ------------------------------------------------------------------ rendering */
typedef struct RtCamera {                       /* arguments of Camera::new, src/cameras.rs:38-48 */
    double look_from[3], look_at[3], v_up[3];
    double vertical_fov_deg, aspect_ratio, aperture, focus_dist, shutter_open, shutter_close;
} RtCamera;
typedef struct RtRenderParams {
    int32_t width, height;      /* src/main.rs:515-516 */
    int32_t spp;                /* samples traced by this call (src/main.rs:570) */
    int32_t sample_offset;      /* absolute index of the first sample: shards of one image use disjoint ranges */
    int32_t max_depth;          /* src/main.rs:580 */
    float background[3];        /* src/main.rs:534 (black in the reference) */
    uint64_t seed;              /* Philox key; (pixel, sample, bounce) is the counter */
    int32_t paths_per_pass;     /* 0 = default; paths in flight per wavefront pass */
    int32_t flags;              /* RT_FLAG_* */
    /* The pixels this call renders -- the reference's outer loops run over all of them (src/main.rs:567-568).  tile = (x0, y0, w, h),
     * a rectangle of the image (row 0 = top row); all zero = the whole image.  stripe = (band_rows, n_shards, shard): of the
     * rectangle's rows keep only the bands of `band_rows` rows that a round-robin deal gives to shard `shard` of `n_shards`; all
     * zero = every row.  The Philox counter holds the pixel's index in the whole image, so the pixels rendered are bit-identical to
     * those of a whole-image call; the others are left untouched.  The buffers are always those of the whole image. */
    int32_t tile[4];
    int32_t stripe[3];
    int32_t reserved_;          /* 0 */
} RtRenderParams;
#define RT_FLAG_SUMSQ 1         /* also accumulate per-channel sum of squares (accum_sq must be given) */
#define RT_FLAG_COUNTERS 2      /* run the instrumented (slower) kernel build and fill RtStats.work */
#define RT_FLAG_TIME_TRACE 4    /* bracket every closest-hit launch with HIP events and fill RtStats.kernel_ms_trace */
#define RT_FLAG_WAVEFRONT_PRIMARY 8 /* rt_trace_primary only: find the hits with the kernels a render of this scene runs at its first
                                     * bounce (camera rays through the ray queue into the scene's closest-hit kernel) instead of the
                                     * plain per-ray search */
#define RT_FLAG_TILES 16        /* rt_render_multi only: shard the PIXELS (row bands of RT_TILE_ROWS rows dealt round-robin), every GPU
                                 * tracing all samples of its pixels, instead of the samples; chosen automatically when spp < n_scenes */
#define RT_TILE_ROWS 8

typedef struct RtStats {
    uint64_t rays;              /* ray segments = executions of src/lib.rs:63 */
    uint64_t paths;             /* width * height * spp */
    uint64_t rays_per_bounce[64]; /* [b-1] = segments traced at bounce b (clamped to 64 entries) */
    double seconds;             /* wall time inside the call (device work, synchronised) */
    double kernel_ms_extend;    /* summed device time of the bounce loops: closest-hit + shade kernels (HIP events) */
    uint64_t extend_launches;   /* bounces launched (one closest-hit and one shade launch each) */
    int32_t passes;
    /* RT_FLAG_COUNTERS only: [0] BVH nodes visited (2 slab tests each), [1] sphere, [2] moving-sphere, [3] rectangle tests,
     * [4] instance entries, [5] medium tests, [6] misses, [7] box tests, [8..12] scatter evaluations per material kind (Lambertian, Metal,
     * Dielectric, DiffuseLight, Isotropic), [13] Noise-textured, [14] Checker-textured shades. */
    uint64_t work[16];
    double kernel_ms_trace;     /* RT_FLAG_TIME_TRACE only: summed device time of the closest-hit kernel alone */
    uint64_t trace_launches;
    /* rt_render_multi only: distinct GPUs that rendered; ranks of the RCCL communicator that carried the reduce (0: no reduce was
     * needed -- one GPU, or pixel sharding); 1 when the pixels were sharded (RT_FLAG_TILES), 0 when the samples were */
    int32_t multi_gpus, multi_comm_ranks, multi_tiles, reserved_;
    /* Of `rays`, those whose closest-hit search (src/lib.rs:63) was settled without the tree: steps of a walk inside a constant
     * medium whose boundary encloses a known handful of surfaces are tested against just those, by the shading kernel that makes
     * the ray (same tests, same operands, same result).  They are ray segments like the others; the closest-hit kernel traced
     * rays - rays_settled. */
    uint64_t rays_settled;
} RtStats;

/* Renders `spp` samples of every pixel and ADDS the per-pixel colour sums (src/main.rs:585-589) to
 * accum_rgb[H][W][3] (row 0 = top image row, as out_image.put_pixel writes it, src/main.rs:592-597).
 * rt_render: host buffers (accum_sq may be NULL).  rt_render_device: device buffers on the scene's
 * GPU, work enqueued on `hip_stream` (a hipStream_t, NULL = default stream) and synchronised
 * before returning so that `stats` is complete. */
int rt_render(RtScene* s, const RtCamera* cam, const RtRenderParams* p, float* accum_rgb, float* accum_sq, RtStats* stats);
int rt_render_device(RtScene* s, const RtCamera* cam, const RtRenderParams* p, float* d_accum_rgb, float* d_accum_sq,
                     void* hip_stream, RtStats* stats);

/* The wavefront workspace (ray queues, hit records, contribution slots: 144 bytes per path of a pass, 38 GB for the default pass of
 * 2^28 paths) belongs to the GPU, not to a scene: all scene handles on one GPU share it, it only grows, and render calls on one GPU
 * take turns on it (calls on different GPUs run side by side).  It is freed when the last scene handle on the GPU is destroyed, or by
 * rt_release_workspace, e.g. before handing the GPU's memory to another library while scenes stay alive; the next render allocates it
 * again.  No counterpart in the reference. */
int rt_release_workspace(int device);

/* rt_render_device for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  It has no counterpart in the reference.  The ray queues, hit records, contribution slots and counters
 * of a pass live in a static 64 MiB device array that every GPU the library runs on holds until the process ends, so the call
 * makes no hipMalloc / hipFree and creates no events.  Limits: width * height <= 262144 (paths of one pass), max_depth <= 126,
 * flags = 0 or RT_FLAG_SUMSQ; anything beyond is RT_ERR_INVALID.  With stats == NULL nothing is read back and the call returns with the work
 * enqueued on `hip_stream` (a call can then be captured into a HIP graph: tests/test_zz_no_alloc.py); with stats the stream is synchronised
 * once per pass.  The queue-overflow check (RT_ERR_INTERNAL) needs that read-back: a caller that passes NULL trusts the capacity bound, which
 * holds by construction for every pass size the entry point accepts.  All calls on one GPU share that GPU's array: calls
 * that overlap on one GPU race (serialise them on one stream).  Results are bit-identical to rt_render_device with
 * paths_per_pass = 262144. */
int rt_render_device_noalloc(RtScene* s, const RtCamera* cam, const RtRenderParams* p, float* d_accum_rgb, float* d_accum_sq,
                             void* hip_stream, RtStats* stats);

/* Multi-GPU render inside one process: the per-pixel sum of src/main.rs:585-589 lifted across GPUs.  scenes[i] is the same world
 * flattened onto a GPU (normally one handle per GPU; handles that share a GPU take turns on it and their sums are added on the
 * device before the reduce).  The samples [sample_offset, sample_offset + spp) are cut into
 * n_scenes contiguous ranges (rt_shard_samples); GPU i renders its range into a private accumulator on its own host thread and
 * stream; one ncclReduce(sum, f32, W*H*3) over RCCL brings them to scenes[0]'s GPU, and the result is ADDED to the host buffers.
 * The Philox counter carries the absolute sample index, so the image equals a one-GPU render up to fp32 summation order; with
 * n_scenes = 1 it is bit-identical to rt_render.  librccl is loaded on first use (RT_ERR_NCCL if absent or if a call fails). */
int rt_render_multi(RtScene* const* scenes, int32_t n_scenes, const RtCamera* cam, const RtRenderParams* p, float* accum_rgb,
                    float* accum_sq, RtStats* stats);
/* The sample range of shard `shard` of `n_shards`: contiguous, balanced, first shards take the remainder. */
int rt_shard_samples(int32_t total_spp, int32_t n_shards, int32_t shard, int32_t* first, int32_t* count);
/* Pixel sharding (the reference's outer loops are over pixels, src/main.rs:567-568): sets p->tile / p->stripe to shard `shard` of
 * `n_shards` of the image p->width x p->height -- row bands of RT_TILE_ROWS rows dealt round-robin, so that every shard sees the
 * whole height of the image -- and returns the shard's pixel count in *n_pixels (may be NULL; 0 when there are fewer bands
 * than shards).  The shards are disjoint and cover the image; a render of each into its own zeroed buffer sums to the whole image
 * bit for bit. */
int rt_shard_tiles(RtRenderParams* p, int32_t n_shards, int32_t shard, int64_t* n_pixels);
/* Destroys the RCCL communicators rt_render_multi has created (they are kept between calls, one set per ordered GPU list).  Call
 * it when no rt_render_multi is running, e.g. before the process ends.  No counterpart in the reference. */
int rt_multi_shutdown(void);

/* Parity hook: closest hit of sample `sample` of every pixel (camera ray + src/lib.rs:63).
 * object_id = creation-order id of the primitive / medium hit, -1 on a miss; t = +inf on a miss.
 * normal (3 per pixel), front_face and uv (2 per pixel) may be NULL. */
int rt_trace_primary(RtScene* s, const RtCamera* cam, const RtRenderParams* p, int32_t sample, int32_t* object_id, float* t,
                     float* normal, int32_t* front_face, float* uv);
/* Parity hook: one scatter event (src/lib.rs:63-65) for n rays given explicitly.  rays: 7 floats
 * each (origin, direction, time); pixel/sample/bounce select the Philox draws of ray i as
 * (pixel0 + i, sample, bounce).  out: 14 floats per ray = status (-1 miss, 0 absorbed, 1 scattered),
 * emitted rgb, attenuation rgb, scattered origin, direction, time. */
int rt_scatter_rays(RtScene* s, const float* rays, int32_t n, uint32_t pixel0, uint32_t sample, int32_t bounce, uint64_t seed,
                    float* out);

/* src/utils.rs:49-63: 256 * clamp(sqrt(sum / spp), 0, 0.999) -> u8, on the host. */
int rt_tonemap_u8(const float* accum_rgb, int32_t n_pixels, int32_t spp, uint8_t* rgb);
/* Writes a binary PPM (P6) of the tonemapped image. */
int rt_write_ppm(const char* path, const uint8_t* rgb, int32_t width, int32_t height);
/* out_image.save(output) of src/main.rs:600: the format follows the file extension as in the `image` crate.  Supported here:
 * .jpeg / .jpg (baseline DCT, 4:4:4, quality 75; ./image.jpeg is the reference's default output, src/main.rs:502), .png (8-bit
 * RGB, stored deflate) and .ppm; anything else is RT_ERR_UNSUPPORTED. */
int rt_write_image(const char* path, const uint8_t* rgb, int32_t width, int32_t height);
/* Checkpoints of the accumulation buffer, so that a long render (configs/final_scene.yml asks for 100 000 spp) can stop and
 * resume: the next call renders with sample_offset = spp_done.  The reference has nothing of the kind (src/main.rs:601 saves
 * once, at the end).  Load fails unless the file holds an image of the given size; *seed (may be NULL) receives the render seed
 * the checkpoint was made with. */
int rt_accum_save(const char* path, const float* accum_rgb, int32_t width, int32_t height, int64_t spp_done, uint64_t seed);
int rt_accum_load(const char* path, float* accum_rgb, int32_t width, int32_t height, int64_t* spp_done, uint64_t* seed);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
