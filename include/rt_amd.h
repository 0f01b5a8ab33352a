/*
 * rt_amd.h — C ABI of the MI355X (gfx950) path-trace backend for the
 * I1Va/RayTracer per-pixel render path.
 *
 * The reference has no FFI of its own: its render entry point is the C++ method
 *   void Camera::render(const SceneManager&, std::pair<int,int>, std::vector<RTPixelColor>&)
 *   (reference inc/Camera.h:64-69, dispatch at src/Camera.cpp:56-67).
 * Everything below sits *under* that method: the host layer flattens a
 * SceneManager into the plain arrays declared here and calls rt_render_host().
 *
 * Conventions
 *   - every function returns 0 on success, a negative rt_status otherwise, and
 *     never throws; rt_last_error() gives a printable reason.  (The reference
 *     uses assert()/silent clamps only, src/Camera.cpp:76-77.)
 *   - one context per device; a context is not re-entrant (calls on one context come from one
 *     host thread at a time; renders are asynchronous, any number may be queued on one stream and
 *     up to 16 may be in flight across streams: one more fails with RT_ERR_STATE).
 *   - plain pointers and sizes only; "d_" arguments are DEVICE pointers valid on
 *     the context's device, "h_" arguments are HOST pointers.
 *   - the library never falls back to a CPU renderer: scenes it cannot render
 *     are rejected with RT_ERR_UNSUPPORTED.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID = -1,     /* bad argument (null pointer, non-positive size, ...) */
    RT_ERR_HIP = -2,         /* a HIP runtime call failed; see rt_last_error()      */
    RT_ERR_UNSUPPORTED = -3, /* feature outside the path this backend renders      */
    RT_ERR_STATE = -4        /* scene or camera not uploaded yet; 16 renders in flight */
} rt_status;

/* Material kinds: the four RTMaterial subclasses, reference inc/RTMaterial.h:63,98,135,200. */
enum {
    RT_MAT_LAMBERTIAN = 0,
    RT_MAT_METAL = 1,
    RT_MAT_DIELECTRIC = 2,
    RT_MAT_EMISSIVE = 3
};

/* One sphere: SphereObject::position_ + radius_ (reference inc/RTObjects.h:31,86).
 * The radius may be negative (hollow-glass trick: the outward normal flips,
 * inc/RTObjects.h:154). */
typedef struct rt_sphere {
    float cx, cy, cz, r;
} rt_sphere;

/* Primitive kinds: the Primitives subclasses the render path can meet
 * (reference inc/RTObjects.h:85 SphereObject, :162 PlaneObject, :323 PolygonObject, :543 CubeObject). */
enum {
    RT_PRIM_SPHERE = 0,
    RT_PRIM_PLANE = 1,
    RT_PRIM_CUBE = 2,
    RT_PRIM_POLYGON = 3
};

/* One primitive of a general scene, 64 bytes.  The array index is the object id: insertion
 * order, which is also the tie-break order of hitClosest.
 *   sphere  : p = centre xyz, radius
 *   plane   : p = point xyz, normal xyz             (PlaneObject::position_, normal_)
 *   cube    : p = centre xyz, half size xyz         (axis-aligned; CubeObject::halfSize_)
 *   polygon : p = unit normal xyz, centroid xyz     (PolygonObject::normal_, position_) and the
 *             vertices first_vertex .. first_vertex + n_vertices - 1 of the vertex array
 *   selected: Primitives::selected_ — the object shows the red selection halo (:96-105) */
typedef struct rt_prim {
    uint32_t type;
    uint32_t mat;
    uint32_t first_vertex;
    uint32_t n_vertices;
    float p[8];
    uint32_t selected;
    uint32_t _pad[3];
} rt_prim;

/* One material, flattened.
 *   albedo   : RTLambertian::diffuse_ / RTMetal::specular_ / RTDielectric::specular_local_
 *              (the colour scatter() returns as attenuation, inc/RTMaterial.h:78,113,147)
 *   fuzz     : RTMetal::fuzz_ (already clamped to <= 1 by the constructor, :103)
 *   ior      : RTDielectric::refractionIndex_ (:137)
 *   emission : RTMaterial::emission_ — honoured for every kind, as emitted() is (:37)
 *   diffuse  : RTMaterial::diffuse_ as read by Light::getDirectLighting
 *              (inc/RTObjects.h:251-252); zero for metal / dielectric / emissive unless the
 *              caller set it through the reference getter.  Only used with lights.
 */
typedef struct rt_material {
    uint32_t type;
    float albedo[3];
    float fuzz;
    float ior;
    float emission[3];
    float diffuse[3];
} rt_material;

/* One Phong point light: the fields of class Light (reference inc/RTObjects.h:220-226). */
typedef struct rt_light {
    float position[3];
    float ambient[3];   /* ambientIntensity_  */
    float diffuse[3];   /* defuseIntensity_   */
    float specular[3];  /* specularIntensity_ (stored by the reference but never read: the
                           specular term is pow(cos, viewLightPow) in white, :309-320) */
    float view_pow;     /* viewLightPow_      */
    float _pad[3];
} rt_light;

/* Camera state the kernels need: Camera::center_ and the three Viewport vectors
 * produced by Camera::updateViewPort (reference src/Camera.cpp:206-221,
 * inc/Camera.h:24-33).  The viewport is always 1x1 at focal length 1. */
typedef struct rt_camera {
    float center[3];
    float upper_left[3];
    float right[3];
    float down[3];
} rt_camera;

/* Render flags */
enum {
    RT_FLAG_ACCUMULATE = 1u << 0, /* add to d_accum instead of overwriting it; the
                                     8-bit image is then made from the running mean */
    RT_FLAG_DIRECT_LIGHT = 1u << 1 /* CameraRenderProperties::enableLDirect: add
                                      Camera::computeDirectLighting at every hit (one shadow
                                      query per uploaded light); inert without lights */
};

/* Render request: CameraRenderProperties (inc/Camera.h:14-22) plus the frame
 * geometry and the strip partition used for multi-GPU rendering.
 *
 * The image is cut into strips of RT_STRIP_ROWS rows.  A call renders strips
 * strip_first, strip_first + strip_stride, ... and stores them *compactly*:
 * local strip k (global strip strip_first + k*strip_stride) occupies rows
 * [k*RT_STRIP_ROWS, (k+1)*RT_STRIP_ROWS) of the output buffers, each row `width`
 * pixels.  With strip_first = 0, strip_stride = 1 the output is the plain
 * row-major width x height image.  Random streams are keyed on the *global*
 * pixel id y*width + x, so any partition produces the same bytes.
 */
#define RT_STRIP_ROWS 8

typedef struct rt_params {
    int32_t width;
    int32_t height;
    int32_t samples_per_pixel; /* CameraRenderProperties::samplesPerPixel              */
    int32_t max_depth;         /* CameraRenderProperties::maxRayDepth                  */
    int32_t sample_offset;     /* index of the first sample (progressive rendering)   */
    uint32_t seed;
    uint32_t flags;
    int32_t strip_first;
    int32_t strip_stride;
} rt_params;

/* Number of rows the compact output of `p` holds (multiple of RT_STRIP_ROWS unless
 * strip_stride == 1, where it is exactly height). */
int rt_output_rows(const rt_params *p);

/* Replaces: nothing in the reference (it has no device); creates the per-device
 * state that Camera::render would otherwise keep on the stack. */
int rt_create(int device, rt_ctx **out);
void rt_destroy(rt_ctx *ctx);
const char *rt_last_error(const rt_ctx *ctx);

/* Replaces SceneManager's primitive list as seen by hitClosest
 * (reference src/RayTracer.cpp:68-82, container inc/RayTracer.h:9,29-32).
 * Sphere order = insertion order = tie-break order.  mat_id[i] indexes mats[]. */
int rt_upload_scene(rt_ctx *ctx, int n_spheres, const rt_sphere *h_spheres,
                    const uint32_t *h_mat_id, int n_mats, const rt_material *h_mats);

/* Same for a scene that also holds planes, axis-aligned cubes, planar polygons or selected
 * objects (Primitives::hit / hitExpanded of each kind, reference inc/RTObjects.h:174-194,
 * 370-386, 455-512, 560-568, 586-644; halo pass src/RayTracer.cpp:84-97).
 * h_vertices holds n_vertices xyz triples (polygon corners, world space); may be NULL when 0. */
int rt_upload_scene_ex(rt_ctx *ctx, int n_prims, const rt_prim *h_prims, int n_vertices,
                       const float *h_vertices, int n_mats, const rt_material *h_mats);

/* Replaces SceneManager's light list as read by Camera::computeDirectLighting
 * (reference src/Camera.cpp:161-178, inc/RayTracer.h:26,30).  n_lights may be 0. */
int rt_upload_lights(rt_ctx *ctx, int n_lights, const rt_light *h_lights);

/* Replaces Camera::center()/viewPort() reads inside genRay (src/Camera.cpp:124-135). */
int rt_set_camera(rt_ctx *ctx, const rt_camera *cam);

/* Replaces Camera::renderParallel (reference src/Camera.cpp:69-88) together with
 * renderPixelColor/genRay/getRayColor/convertRTColor (:105-157, :19-34) and
 * SceneManager::hitClosest / SphereObject::hit / RTMaterial::scatter beneath them.
 *   d_rgba  : device, rt_output_rows()*width RGBA8 pixels (RTPixelColor, inc/Camera.h:10-12)
 *   d_accum : device, optional (may be NULL), rt_output_rows()*width float4
 *             (sum r, sum g, sum b, number of samples)
 *   d_rays  : device, optional, one uint64 that the call ADDS the number of
 *             closest-hit queries to
 *   stream  : a hipStream_t (NULL = the default stream).  The call is asynchronous.
 * Any number of renders may be queued on one stream; across streams up to 16 renders of one context may be in
 * flight, and one more on yet another stream fails with RT_ERR_STATE and launches nothing.
 */
int rt_render(rt_ctx *ctx, const rt_params *p, void *d_rgba, void *d_accum,
              uint64_t *d_rays, void *stream);

/* No counterpart in the reference.  For callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc): the same render as
 * rt_render() with the kernel's work counter taken from a device array of the library's code object
 * (RT_NOALLOC_SCRATCH_BYTES per GPU, held until the process ends) instead of the context's
 * allocation.  The call itself allocates and frees nothing; scene, camera and the output buffers
 * are the caller's and the context's, set up beforehand as for rt_render().
 * Limits: at most 2^31 - 1 work items per call (width*height < 2^31), rejected with RT_ERR_INVALID
 * beyond that.  All contexts on one GPU share the array's 16 counters; each is handed out again once the
 * launch that used it has finished (an event per counter, created at rt_create) or to a call on the stream that
 * launch ran on: any number of calls may be queued on one stream, up to 16 may be in flight on one GPU across
 * streams, and a call beyond that fails with RT_ERR_STATE and launches nothing.  On a stream under
 * graph capture the events cannot be asked, and the counters are used round robin unchecked. */
#define RT_NOALLOC_SCRATCH_BYTES 4096
int rt_render_noalloc(rt_ctx *ctx, const rt_params *p, void *d_rgba, void *d_accum,
                      uint64_t *d_rays, void *stream);

/* Same, with HOST outputs and a synchronous copy-out: this is the function a
 * Camera::render binding calls with outputBufer.data().  h_rays receives (is
 * overwritten with) the number of closest-hit queries. */
int rt_render_host(rt_ctx *ctx, const rt_params *p, uint8_t *h_rgba, float *h_accum,
                   uint64_t *h_rays);

/* Duration in milliseconds of the render kernel of the last rt_render_host() call (HIP events on the
 * context's stream), and the number of device allocations (hipMalloc calls) made for the context so
 * far: its buffers are kept and only grown, so a second frame of an unchanged scene adds none.
 * Replaces nothing in the reference; for callers that drive the camera per frame
 * (Camera::move / rotate then render, reference src/Camera.cpp:43-67). */
float rt_last_kernel_ms(const rt_ctx *ctx);
uint64_t rt_device_allocations(const rt_ctx *ctx);

/* Which form of the render kernel the context's last render launched (-1: none yet): where its sphere scan takes
 * the spheres from -- 0 their LDS-resident copy, 1 tiles streamed through LDS, 2 scalar loads (the packed FP32
 * filter), 3 the matrix-core filter with its table resident in LDS, 4 the same with the table streamed through
 * LDS -- plus 8 when it was the general kernel (lights, other primitive kinds, the selection halo).  Replaces
 * nothing in the reference; for tests and tuning. */
int rt_last_kernel_variant(const rt_ctx *ctx);

/* Scans the waves of the last rt_render_host() call's kernel ran: one scan finds the closest sphere for the
 * (up to) 64 rays a wave holds, so 64 x this x the table rows of rt_filter_describe() is the number of filter
 * values the matrix cores produced.  Primary rays decided from their tile's list of spheres take no scan.
 * Replaces nothing in the reference; for measurement. */
uint64_t rt_last_scans(const rt_ctx *ctx);

/* Table blocks (32 rows against the 64 rays of a wave, four matrix instructions) the scans of the last
 * rt_render_host() call walked: scans x blocks of the table, fewer when the kernel walks a long table only where a
 * wave's rays can meet a block's bounding sphere (variant 5; the blocks of the table of bounds count too).
 * Replaces nothing in the reference; for measurement. */
uint64_t rt_last_blocks(const rt_ctx *ctx);

/* First-hit flags */
enum {
    RT_FH_REFINE = 1u << 0,  /* re-test numerically ambiguous rays in fp64 so that the
                                returned id equals a double-precision evaluation */
    RT_FH_EXPANDED = 1u << 1 /* hitExpandedState = true: add the selection-halo pass
                                (src/RayTracer.cpp:84-97) */
};

/* Replaces SceneManager::hitClosest(ray, Interval(tmin, +inf), rec, false)
 * (reference src/RayTracer.cpp:68-82) for a batch of rays.
 *   org, dir : n_rays x 3 floats (directions need not be unit length,
 *              inc/RTObjects.h:136)
 *   id       : index of the closest sphere or -1;  t: HitRecord::time;
 *   normal   : HitRecord::normal (unit, facing against the ray);
 *   front    : HitRecord::frontFace (0/1).  t/normal/front may be NULL.
 *   tmin     : any float; with tmin <= 0 roots behind the origin count, as Interval::surrounds
 *              (inc/RTGeometry.h:79-81) has it.
 * Host-pointer version (copies in and out, synchronous). */
int rt_first_hit_host(rt_ctx *ctx, int n_rays, const float *h_org, const float *h_dir,
                      float tmin, uint32_t flags, int32_t *h_id, float *h_t,
                      float *h_normal, uint8_t *h_front);

/* Device-pointer version, asynchronous on `stream`. */
int rt_first_hit(rt_ctx *ctx, int n_rays, const float *d_org, const float *d_dir,
                 float tmin, uint32_t flags, int32_t *d_id, float *d_t, float *d_normal,
                 uint8_t *d_front, void *stream);

/* The same two calls with HitRecord::hitExpanded as one more output (0/1, may be NULL): set when
 * the ray met the grown shell of a selected object before anything else.  For a general scene
 * `id` is the index into the rt_prim array. */
int rt_first_hit_ex_host(rt_ctx *ctx, int n_rays, const float *h_org, const float *h_dir,
                         float tmin, uint32_t flags, int32_t *h_id, float *h_t,
                         float *h_normal, uint8_t *h_front, uint8_t *h_expanded);
int rt_first_hit_ex(rt_ctx *ctx, int n_rays, const float *d_org, const float *d_dir,
                    float tmin, uint32_t flags, int32_t *d_id, float *d_t, float *d_normal,
                    uint8_t *d_front, uint8_t *d_expanded, void *stream);

/* Replaces RTMaterial::scatter (reference inc/RTMaterial.h:67-80,105-115,142-170,
 * 205-207) for a batch of hits, with the three uniform variates the bounce would
 * draw passed in explicitly (u[0],u[1] -> IVec3f::randomUnit, u[2] -> randomDouble).
 * Host pointers, synchronous.  All arrays are n x 3 floats except mat (n),
 * front (n) and scattered_flag (n).  Used by parity tests of the bounce step. */
int rt_scatter_host(rt_ctx *ctx, int n, const uint32_t *h_mat, const float *h_in_dir,
                    const float *h_point, const float *h_normal, const uint8_t *h_front,
                    const float *h_u, float *h_att, float *h_out_org, float *h_out_dir,
                    uint8_t *h_scattered_flag);

/* ---- device groups: one frame on several GPUs of one process ---------------------------------
 * Replaces the OpenMP team of Camera::renderParallel (reference src/Camera.cpp:69-88: pixels are
 * independent and each is seeded by its own id) by a team of GPUs: member i of n renders the 8-row
 * strips i, i+n, ... on its own stream, the strips are gathered on the first member's GPU (RCCL
 * send/recv over xGMI between distinct GPUs, device copies when a GPU is listed more than once)
 * and dealt back into row order there.  The frame equals the one-GPU frame byte for byte.
 * A group owns one context per list entry; scene, lights and camera are replicated to all. */
typedef struct rt_group rt_group;

typedef struct rt_group_timing {
    float kernel_ms_min; /* shortest and longest render kernel over the members (HIP events)  */
    float kernel_ms_max;
    float gather_ms;     /* total_ms - kernel_ms_max: launches, exchange, assembly, copy-out   */
    float total_ms;      /* host wall time of the call                                          */
} rt_group_timing;

int rt_group_create(int n_members, const int *devices, rt_group **out);
void rt_group_destroy(rt_group *g);
const char *rt_group_last_error(const rt_group *g);
int rt_group_size(const rt_group *g);
const char *rt_group_transport(const rt_group *g); /* "rccl", "copy" or "rccl-loopback" */
uint64_t rt_group_device_allocations(const rt_group *g);
int rt_group_upload_scene_ex(rt_group *g, int n_prims, const rt_prim *h_prims, int n_vertices,
                             const float *h_vertices, int n_mats, const rt_material *h_mats);
int rt_group_upload_lights(rt_group *g, int n_lights, const rt_light *h_lights);
int rt_group_set_camera(rt_group *g, const rt_camera *cam);
/* p->strip_first / strip_stride must be 0 / 1 (the group deals the strips).  *d_frame receives the
 * assembled frame in the first member's device memory (height rounded up to whole strips, valid
 * until the next call); h_rays the closest-hit queries of all members; timing may be NULL.
 * Synchronous. */
int rt_group_render(rt_group *g, const rt_params *p, void **d_frame, uint64_t *h_rays,
                    rt_group_timing *timing);
/* Same with the width*height*4 bytes of the frame copied to host memory: what a multi-GPU
 * Camera::render binding calls with outputBufer.data(). */
int rt_group_render_host(rt_group *g, const rt_params *p, uint8_t *h_rgba, uint64_t *h_rays,
                         rt_group_timing *timing);

/* The counter-based generator the kernels use, exposed for tests:
 * fills out[4] with the four 32-bit words of stream (pixel, sample, call, seed). */
void rt_rng4(uint32_t pixel, uint32_t sample, uint32_t call, uint32_t seed, uint32_t out[4]);
/* The same generator evaluated ON THE DEVICE, by the code the render kernel runs, for n keys of four
 * words (pixel, sample, call, seed) each; h_out receives four words per key.  Host pointers,
 * synchronous.  Stands where the reference seeds its per-thread generator per pixel and draws from it
 * (src/Camera.cpp:84,129-130; inc/RTMaterial.h:72,111,158). */
int rt_rng4_device_host(rt_ctx *ctx, int n, const uint32_t *h_keys, uint32_t *h_out);

/* The sphere filter of the matrix-core scan, exposed for tests.  Scenes of 16 spheres and more are scanned by
 * a conservative filter evaluated on the matrix cores (every sphere a ray could hit passes; what passes is decided exactly), in a frame centred on the sphere cloud; spheres much
 * larger than the cloud are kept out of it and decided directly.  No counterpart in the reference: it is
 * SceneManager::hitClosest's loop over all objects (src/RayTracer.cpp:77-82) in another arithmetic. */
typedef struct rt_filter_info {
    int32_t in_use;     /* 1 if renders of the uploaded scene (sphere-only, no lights) use this filter */
    float center[3];    /* centre of the frame */
    float radius;       /* its unit length: the radius of the sphere cloud */
    float margin;       /* a sphere passes when disc / radius^2 + margin + margin_far * max|O'_i| > 0 */
    float margin_far;   /*   (disc: discriminant of the unit-direction quadratic; O': origin in frame units) */
    int32_t n_rows;     /* columns of rt_filter_values_host: one per sphere index, padded to a multiple of 32 */
    int32_t n_big;      /* spheres decided outside the filter ... */
    int32_t big[8];     /* ... and their indices; their columns never pass */
    int32_t n_table_rows; /* rows of the table the kernel multiplies (the other spheres in the leaf order of a k-d
                           * partition, padded to blocks of 32 and, when too long for LDS, to tiles of 20 blocks) */
} rt_filter_info;
int rt_filter_describe(rt_ctx *ctx, rt_filter_info *out);
/* Filter values of n rays (origin, direction: 3 floats each, direction of any length) against every sphere,
 * computed by the device code the render kernel runs: values[ray * n_rows + sphere index], in frame units with
 * the sign convention of the kernel (NEGATIVE = the sphere passes), i.e. an approximation of
 * -(disc / radius^2 + margin + margin_far * max|O'_i|).  Host pointers, synchronous. */
int rt_filter_values_host(rt_ctx *ctx, int n_rays, const float *h_org, const float *h_dir, float *h_values);

/* The filter's tables for a list of spheres, built on the HOST exactly as an upload builds them (no context, no
 * GPU): for tests that check the tables themselves.  A table too long for LDS gets a table of bounds in front of it:
 * one sphere around the 32 spheres of each table block, in a frame of its own (the cloud's thin axes stretched), and
 * per bound the row of the test for what lies behind a ray; a wave then reads only the blocks whose bound some ray of
 * it can meet.  `info` is always filled; each array may be NULL, else it receives
 *   row_index    : n_rows ints, the sphere index behind every table row (-1: padding),
 *   table_words  : (n_blocks + 1) * 512 words, the table in the lane layout of the matrix instruction's operand,
 *   bound_words  : (n_bound_blocks + 1) * 768 words, the table of bounds likewise (three k-steps per block of 32).
 * Returns RT_ERR_UNSUPPORTED for a list that gets no table.  No counterpart in the reference (its hitClosest asks
 * every object, src/RayTracer.cpp:77-82). */
typedef struct rt_filter_tables_info {
    float center[3];      /* centre of both frames */
    float scale;          /* 1024 / radius of the sphere cloud: table frame units (x 1024) per world unit */
    float bound_scale[3]; /* the same per axis for the frame of the bounds */
    int32_t n_rows;       /* rows of the table, padding included: 32 * (n_blocks + 1) */
    int32_t n_blocks;     /* table blocks of 32 rows (padded to whole tiles of 20 when the table is too long for LDS) */
    int32_t n_used;       /* ... of which hold a sphere */
    int32_t n_bound_blocks; /* blocks of 32 bounds (0: the table stays resident in LDS and has none) */
    int32_t n_big;        /* spheres decided outside the table ... */
    int32_t big[8];       /* ... and their indices */
} rt_filter_tables_info;
int rt_filter_tables_host(int n_spheres, const rt_sphere *h_spheres, rt_filter_tables_info *info, int32_t *row_index,
                          uint32_t *table_words, uint32_t *bound_words);

/* Library/ABI version (major*100 + minor). */
int rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
