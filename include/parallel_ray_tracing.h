/*
 * parallel_ray_tracing.h - C-ABI of libparallel_ray_tracing.so (MI355X / gfx950 build)
 *
 * Drop-in boundary for photon's ray-tracing core.  photon's Python driver loads the
 * library with ctypes (python_codes/perform_ray_tracing_03.py:1888) and binds exactly
 * one symbol, `start_ray_tracing` (argtypes at :1914-1921, restype None at :1925).
 * The struct layouts below are the wire format of that call: they must agree byte for
 * byte with the ctypes.Structure classes the reference builds at
 * perform_ray_tracing_03.py:1651-1659 (scattering), :1708-1720 (source),
 * :1751-1786 (element), :1838-1852 (camera), which in turn mirror
 * cuda_codes/parallel_ray_tracing.h:17-191.  tests/test_abi.py checks sizeof/offsetof
 * against fixtures captured from the reference's marshalling code.
 *
 * Everything is plain C: pointers, sizes, PODs.  No torch / HIP types appear in a
 * signature (streams and device pointers travel as void*).
 *
 * Section 1  = the reference's ABI (what photon binds today).
 * Section 2  = `photon_*` extension entry points (device-resident scene / volume /
 *              image handles) that bench.py and the parity tests bind; the reference
 *              has no counterpart, each cites the part of start_ray_tracing it factors out.
 */
#ifndef PHOTON_AMD_PARALLEL_RAY_TRACING_H_
#define PHOTON_AMD_PARALLEL_RAY_TRACING_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------
 * Section 1: wire structs (x86-64 SysV natural alignment; sizes in bytes in brackets)
 * ---------------------------------------------------------------------------------- */

/* Mie-scattering lookup data [72].  Replaces cuda_codes/parallel_ray_tracing.h:17-33.
 * scattering_irradiance is row-major [num_angles][num_diameters]; both pointers are
 * NULL and the float fields NaN when scattering_type_str != "mie"
 * (perform_ray_tracing_03.py:1685-1701). */
typedef struct scattering_data_t {
    float inverse_rotation_matrix[9];   /* @0  camera -> world */
    float beam_propagation_vector[3];   /* @36 unit vector of the laser sheet */
    float *scattering_angle;            /* @48 [num_angles], radians, uniform spacing */
    float *scattering_irradiance;       /* @56 [num_angles*num_diameters] */
    int num_angles;                     /* @64 */
    int num_diameters;                  /* @68 */
} scattering_data_t;

/* Light-field sources = particles / dot-pattern points [64].
 * Replaces cuda_codes/parallel_ray_tracing.h:36-60. */
typedef struct lightfield_source_t {
    int lightray_number_per_particle;   /* @0  */
    int source_point_number;            /* @4  sources per launch chunk (10000 in photon) */
    int *diameter_index;                /* @8  [num_particles] column of the Mie table */
    double *radiance;                   /* @16 [num_particles] */
    float *x;                           /* @24 [num_particles] microns, camera frame */
    float *y;                           /* @32 */
    float *z;                           /* @40 */
    int num_particles;                  /* @48 */
    float z_offset;                     /* @52 z_object - object_distance */
    float object_distance;              /* @56 */
} lightfield_source_t;

/* [32] cuda_codes/parallel_ray_tracing.h:115-131 */
typedef struct element_geometry_t {
    float front_surface_radius;         /* @0  */
    bool front_surface_spherical;       /* @4  */
    float back_surface_radius;          /* @8  */
    bool back_surface_spherical;        /* @12 */
    float pitch;                        /* @16 clear aperture diameter */
    double vertex_distance;             /* @24 centre thickness */
} element_geometry_t;

/* [24] cuda_codes/parallel_ray_tracing.h:134-141 */
typedef struct element_properties_t {
    float abbe_number;                  /* @0  NaN = no dispersion */
    float absorbance_rate;              /* @4  */
    double refractive_index;            /* @8  */
    float thin_lens_focal_length;       /* @16 */
    float transmission_ratio;           /* @20 */
} element_properties_t;

/* One optical element [120].  cuda_codes/parallel_ray_tracing.h:144-162.
 * element_type: 'l' thick spherical lens, 't' thin lens, 'n' apparent image (no lens),
 * anything else = aperture stop (parallel_ray_tracing.cu:416,507,868; :2143). */
typedef struct element_data_t {
    double axial_offset_distances[2];   /* @0  */
    element_geometry_t element_geometry;/* @16 */
    float element_number;               /* @48 */
    element_properties_t element_properties; /* @56 */
    char element_type;                  /* @80 */
    float elements_coplanar;            /* @84 */
    double rotation_angles[3];          /* @88 */
    float z_inter_element_distance;     /* @112 */
} element_data_t;

/* Sensor description [112].  cuda_codes/parallel_ray_tracing.h:165-191. */
typedef struct camera_design_t {
    int pixel_bit_depth;                /* @0  */
    float pixel_gain;                   /* @4  */
    float pixel_pitch;                  /* @8  microns */
    float x_camera_angle;               /* @12 */
    float y_camera_angle;               /* @16 */
    int x_pixel_number;                 /* @20 image width  W */
    int y_pixel_number;                 /* @24 image height H */
    float z_sensor;                     /* @28 */
    float diffraction_diameter;         /* @32 pixels */
    bool implement_diffraction;         /* @36 true: erf splat, false: 4-pixel splat */
    float rotation_matrix[9];           /* @40 world -> camera */
    float inverse_rotation_matrix[9];   /* @76 camera -> world */
} camera_design_t;

/*
 * start_ray_tracing - render one sensor image.
 * Replaces cuda_codes/parallel_ray_tracing.cu:3078-3775 (declared
 * cuda_codes/parallel_ray_tracing.h:303-307).  Argument meaning is unchanged:
 *   image_array        f32[H*W], row-major row*W+col, READ-MODIFY-WRITE (accumulates on
 *                      the caller's contents, .cu:3309,3675)
 *   element_center     f64[num_elements][3]; element_plane_parameters f64[num_elements][4]
 *   ray_tracing_algorithm  1 euler, 2 rk4, 3 rk45, 4 adams-bashforth; any other value leaves the ray
 *                      straight (the reference's `default: break`, trace_rays_...h:1537).  3 and 4
 *                      are restated literally, trilinear on the raw volume whatever PHOTON_INTERP
 *   density_grad_filename  NRRD (type float, dimension 3; encoding raw, gzip or ascii; either byte order), "" when unused
 *   save_lightrays     writes <pos_path>/pos_%04d.bin, <dir_path>/dir_%04d.bin per chunk
 * All pointers are borrowed for the duration of the call.  No error channel (void):
 * on failure a message goes to stderr and image_array is left unmodified.
 * Environment knobs (the ABI has no room for new arguments):
 *   PHOTON_INTERP=linear|cubic   volume sampler (default linear = the reference's
 *                                hard-coded interpolation_scheme 1, .cu:3330)
 *   PHOTON_TEX_WEIGHTS=fixed8|exact   trilinear weights: 8 fractional bits like the texture unit the reference's
 *                                tex3D() runs on (default) or exact f32 (photon_volume_set_weight_bits)
 *   PHOTON_VERBOSE=1             progress / timing on stdout
 *   PHOTON_NOISE_SEED=u64        seed of the add_pos_noise / add_ngrad_noise generators (the
 *                                reference seeds cuRAND with time(NULL); default 0x5eed)
 *   PHOTON_ELEMENT_TRAIN=reference|sequential   element-group walk (photon_scene_set_element_train)
 *   PHOTON_SKIP_DOOMED=0|1       1 (default): rays that provably die on the first aperture are not marched; without a volume
 *                                the lens samples no source can get through it and the sources whose image cannot fall on the
 *                                sensor are not launched (photon_scene_set_skip_doomed); the image is the same bit for bit
 *   PHOTON_RAYGEN=fold|kernel    where the rays of a launch through a volume are generated: in the prologue of the march's
 *                                first piece (default) or by a kernel of its own; same bits (read once per process)
 *   PHOTON_RAY_ORDER=source|lens|auto           lane order of a launch (photon_scene_set_ray_order)
 *   PHOTON_DEVICES=all|0,1,..    shard the sources of one call over several GPUs: one host thread per
 *                                device uploads only its block of the sources, the NRRD is parsed once, the
 *                                private f64 accumulators are summed on the first device (peer copies +
 *                                a kernel) and folded into image_array once; default: the calling
 *                                thread's current device (which is restored on return in every case)
 */
void start_ray_tracing(float lens_pitch, float image_distance,
                       scattering_data_t *scattering_data_p, char *scattering_type_str,
                       lightfield_source_t *lightfield_source_p,
                       int lightray_number_per_particle, float beam_wavelength,
                       float aperture_f_number, int num_elements,
                       double (*element_center)[3], element_data_t *element_data_p,
                       double (*element_plane_parameters)[4], int *element_system_index,
                       camera_design_t *camera_design_p, float *image_array,
                       bool simulate_density_gradients, char *density_grad_filename,
                       bool save_lightrays, char *lightray_position_save_path,
                       char *lightray_direction_save_path, int num_lightrays_save,
                       int ray_tracing_algorithm, bool add_pos_noise, float pos_noise_std,
                       bool add_ngrad_noise, float ngrad_noise_std,
                       float ray_cone_pitch_ratio, bool save_intermediate_ray_data,
                       int num_intermediate_positions_save);

/* ------------------------------------------------------------------------------------
 * Section 2: device-resident extension API (what bench.py / tests bind)
 * Return value: 0 on success, non-zero HIP / argument error (message on stderr).
 * ---------------------------------------------------------------------------------- */

typedef struct photon_volume photon_volume_t;   /* refractive-index gradient volume in HBM */
typedef struct photon_scene photon_scene_t;     /* sources, tables, optics, camera in HBM  */

/* Host-visible description of a loaded volume (mirrors density_grad_params_t,
 * cuda_codes/parallel_ray_tracing.h:213-252, minus the pointers). */
typedef struct photon_volume_info_t {
    float min_bound[3];
    float max_bound[3];
    int nx, ny, nz;
    float grid_spacing[3];
    float step_size;
    float data_min;         /* min over the volume of n-1 */
    int interpolation;      /* 1 trilinear, 2 tricubic B-spline */
} photon_volume_info_t;

/* Per-trace counters (filled from device atomics; for roofline accounting).  LAYOUT FROZEN at 72 bytes (round 3): the
 * library writes sizeof(photon_trace_stats_t) bytes through the caller's pointer, so the struct never grows again -- later
 * measurements come through structs that carry their own size (photon_march_profile_t). */
typedef struct photon_trace_stats_t {
    uint64_t rays_launched;
    uint64_t rays_on_sensor;        /* rays that reached the splat stage inside the sensor */
    uint64_t rk_iterations;         /* completed integrator iterations, summed over rays */
    uint64_t volume_samples;        /* sampler invocations, summed over rays */
    uint64_t sensor_taps;           /* atomic adds issued */
    float march_ms;                 /* HIP-event time of the volume-march kernel(s) */
    float total_ms;                 /* HIP-event time of the whole trace */
    uint64_t rays_marched;          /* rays that entered the volume march: rays_launched minus those dropped before it
                                       because they provably die on the first aperture (photon_scene_set_skip_doomed);
                                       0 without a volume */
    float shader_clock_mhz;         /* clock the march kernel actually ran at: s_memtime / s_memrealtime ticks summed over its
                                       waves x 100 MHz (the chip lowers its clock under load, differently from device to
                                       device); 0 without a volume */
    uint32_t traces;                /* photon_trace calls these numbers cover (1, or the calls of a statistics window) */
    float march_wave_ms;            /* mean time a march wave spends on one 64-ray group, from the same stamps: a launch of G
                                       groups lasts about G / (waves resident on the chip) of these */
} photon_trace_stats_t;

/* Select the GPU this thread's subsequent photon_* calls use (hipSetDevice). */
int photon_set_device(int device);

/* PCI bus id ("0000:c1:00.0") of the calling thread's current device: names its sysfs node
 * (/sys/bus/pci/devices/<id>/hwmon/...: board power, cap) for measurement scripts. */
int photon_device_pci_bus_id(char *buf, int len);

/* glibc-compatible lens-sample table, factored out of parallel_ray_tracing.cu:3216-3243
 * (srand(10); r1[k]=rand()/RAND_MAX; r2[k]=rand()/RAND_MAX, interleaved).  Host arrays. */
int photon_rand_table(int n, float *r1, float *r2);

/* NRRD load + n-1 / grad(n) volume build (+ B-spline prefilter when interpolation==2).
 * Replaces readDatafromFile/loadNRRD/setData/Host_Init,
 * cuda_codes/trace_rays_through_density_gradients.h:1612-2105. */
int photon_volume_load_nrrd(const char *path, int interpolation, photon_volume_t **out);
/* Same, from a density field already on the host (x fastest), for synthetic volumes.
 * origin is the NRRD "space origin" BEFORE the reference's -750e3 z shift (.h:1704). */
int photon_volume_from_density(const float *rho, int nx, int ny, int nz,
                               const double spacing[3], const double origin[3],
                               int interpolation, photon_volume_t **out);
int photon_volume_info(const photon_volume_t *vol, photon_volume_info_t *info);
/* Trilinear interpolation weights: bits = 8 (default) rounds them to 8 fractional bits, the 9-bit fixed-point
 * weights CUDA's linear texture filter uses (CUDA C Programming Guide, "Linear Filtering") -- i.e. what the
 * reference's tex3D() fetches (trace_rays_through_density_gradients.h:1052 ...) compute with on NVIDIA
 * hardware; bits = 0 keeps them exact f32.  Takes effect for every later sample / trace of this volume; the
 * tricubic sampler (always the exact 64-tap sum) is unaffected.  start_ray_tracing reads
 * PHOTON_TEX_WEIGHTS=fixed8|exact. */
int photon_volume_set_weight_bits(photon_volume_t *vol, int bits);
/* Copy the float4 texels (grad x,y,z, n-1) -- or the B-spline coefficients when
 * interpolation==2 and coefficients!=0 -- back to the host: f32[nz*ny*nx*4]. */
int photon_volume_download(const photon_volume_t *vol, int coefficients, float *out);
/* Sample at n unnormalised texel coordinates (the argument of tex3D / cubicTex3D,
 * trace_rays_through_density_gradients.h:1052,1216): coords f32[n][3] -> out f32[n][4].
 * Host arrays; used by the sampler parity tests. */
int photon_volume_sample(const photon_volume_t *vol, int n, const float *coords, float *out);
void photon_volume_free(photon_volume_t *vol);

/* Upload everything start_ray_tracing copies to the device before its launch loop
 * (parallel_ray_tracing.cu:3132-3314): same arguments, same meaning. */
int photon_scene_create(float lens_pitch, float image_distance,
                        const scattering_data_t *scattering_data_p,
                        const char *scattering_type_str,
                        const lightfield_source_t *lightfield_source_p,
                        int lightray_number_per_particle, float beam_wavelength,
                        float aperture_f_number, int num_elements,
                        const double (*element_center)[3], const element_data_t *element_data_p,
                        const double (*element_plane_parameters)[4],
                        const int *element_system_index,
                        const camera_design_t *camera_design_p, float ray_cone_pitch_ratio,
                        photon_scene_t **out);
/* Waits for the scene's device first when a trace of this scene may still be running (its device blocks go back to the
 * library's block cache, see photon_trim_caches, and may be handed to the next scene at once): freeing a scene right after an
 * asynchronous photon_trace is safe on any stream.  A scene belongs to the device that was current when it was created;
 * photon_trace, photon_scene_free and the statistics calls make that device current for their duration and restore the
 * caller's, so they may be called with any device current. */
void photon_scene_free(photon_scene_t *scene);

/* Noise hooks of start_ray_tracing (its add_pos_noise / pos_noise_std / add_ngrad_noise /
 * ngrad_noise_std arguments, parallel_ray_tracing.cu:3405-3445): Gaussian jitter of the sensor hit
 * (sigma in pixels) and of dn/dx, dn/dy in the Euler march.  Counter-based generator keyed by
 * `seed` (include/photon_philox.h); off by default. */
int photon_scene_set_noise(photon_scene_t *scene, int add_pos_noise, float pos_noise_std, int add_ngrad_noise,
                           float ngrad_noise_std, uint64_t seed);

/* How element groups are walked (propagate_rays_through_optical_system, parallel_ray_tracing.cu:1274-1381).
 * mode 0 (default) = the reference as it runs: every single-member group goes through element 0
 * (:1331-1333), groups of simultaneous elements reach an empty stub (:1049-1272).  mode 1 = the working
 * train: each group through its own element(s), simultaneous elements (lenslet arrays, any number)
 * chosen per ray by nearest centre on the element plane (design: perform_ray_tracing_03.py:1254-1485).
 * start_ray_tracing reads it from PHOTON_ELEMENT_TRAIN=reference|sequential. */
int photon_scene_set_element_train(photon_scene_t *scene, int mode);

/* The partition of a march launch over its work queues (host restatement of the kernel's own functions, for tests).
 * photon_march_queue_count() queues -- 8 XCDs x (count / 8) sub-queues, 32 in the shipped build; consecutive groups form
 * CHUNKS of groups (photon_march_queue_chunk(interpolation): 16 for the tricubic kernels in source-major launches through
 * volumes of up to 256^3 texels, 128 for the trilinear kernels -- and for every lens-major launch or larger volume), and
 * queue (xcd, sub) owns the chunks c with c % count == sub * 8 + xcd.  photon_march_queue_group: index of the k-th 64-ray
 * group that queue hands out (grows with k); photon_march_queue_size: how many of a launch's n_groups groups it owns.
 * Every group of a launch belongs to exactly one queue.  An xcd >= 8, a sub >= count / 8 or a groups_per_chunk that is
 * not a power of two up to 65536 returns UINT_MAX.
 * A launch whose marches are cut into S segments (photon_scene_set_march_segments) hands out size * S items per queue,
 * segment-major: item k is segment k / size of the queue's (k % size)-th group. */
unsigned photon_march_queue_count(void);
unsigned photon_march_queue_chunk(int interpolation);
unsigned photon_march_queue_group(unsigned k, unsigned xcd, unsigned sub, unsigned groups_per_chunk);
unsigned photon_march_queue_size(unsigned n_groups, unsigned xcd, unsigned sub, unsigned groups_per_chunk);

/* Segments per march (speed only; the image does not depend on it, nor do the marched rays: tests).  A 64-ray group marches
 * for ~2 ms and a launch ends when its last group does, so the chip idles for most of a group time at the end of every
 * launch; a launch of several chip fills therefore cuts every march into `segments` pieces of equal trip count, handed out
 * breadth-first, and a ray's loop state travels with it from piece to piece.  -1 (default) = the library's choice
 * (at most PHOTON_MARCH_SEGMENTS, 32, in launches of at least 1.25 chip fills: more pieces the shorter the launch and the
 * longer a march), 1 = whole marches, 2..64 = that many in every
 * launch, whatever its size (tests).  Launches that write intermediate ray dumps or use gradient noise are never segmented.
 * start_ray_tracing reads PHOTON_MARCH_SEGMENTS=<n> (the library's choice) or force:<n> (every launch). */
int photon_scene_set_march_segments(photon_scene_t *scene, int segments);
/* The library's own choice for a launch of n_rays through a volume whose longest axis has `depth` texels, on a device of
 * num_cus compute units: returns the number of pieces (1 = whole marches), *halving (may be NULL) = 1 when their lengths
 * halve (1/2, 1/4, ... of the depth) rather than being equal.  Pure host arithmetic (a cost model fitted to a sweep on C3:
 * DESIGN.md section 4.1); for tests and documentation. */
int photon_march_segments_plan(unsigned n_rays, int depth, int ray_tracing_algorithm, int interpolation, int num_cus, int *halving);

/* A scene that holds only a SLICE of a job's source list (one rank of a multi-GPU job uploads just its shard): the
 * index, in the job's list, of this scene's first source.  Only the noise hooks read it -- their generator is keyed by the
 * ray's place in the whole job, so a sharded render draws the numbers the unsharded one draws.  Default 0. */
int photon_scene_set_source_base(photon_scene_t *scene, int64_t first_source);

/* Order in which a launch lays its rays over the GPU's lanes (results are a sum: the image does not depend
 * on it beyond f64 summation order).  0 = source-major, the reference's thread order (.cu:1988-2006): best
 * when a source's ray cone is narrower than a volume texel (BOS).  1 = lens-major over spatially sorted
 * sources: a wave carries 64 neighbouring sources aimed at one lens point -- best when the cone is as wide as
 * the aperture (PIV through a volume).  2 = choose per launch from the cone width at the volume and the
 * texel size (default).  Launches that write ray dumps or use gradient noise are always source-major; with
 * mode 1 or 2 the [src_begin, src_end) of photon_trace counts sources in the sorted order.
 * start_ray_tracing reads PHOTON_RAY_ORDER=source|lens|auto. */
int photon_scene_set_ray_order(photon_scene_t *scene, int mode);

/* Rays that cannot reach the sensor need not be marched (default on).  The reference kills a ray that meets the
 * first element's front surface more than pitch/2 from the axis (thin lens .cu:447, thick lens .cu:560-566) --
 * half of a full-aperture cone, whose lens samples reach out to a radius of one pitch (.cu:123-124) -- but only
 * after marching it through the volume.  With the switch on, a ray whose UNDEFLECTED path misses the aperture by
 * more than the largest footprint shift the volume can cause (bounded from the volume's largest |grad n|) is dropped
 * before the march: same image, same rays_on_sensor; rk_iterations / volume_samples count only the rays marched.
 * Off automatically for launches that write ray dumps, use gradient noise, integrators 3 / 4 or the element
 * train.  start_ray_tracing reads PHOTON_SKIP_DOOMED=0|1.
 * WITHOUT a volume the same switch keeps the dead LENS SAMPLES from being launched at all: ray k of every source is aimed at
 * the same point of the lens plane (.cu:123-141), so which samples miss the aperture is decided once per scene, from the
 * caller's source arrays, with a bound that holds for every source (photon_scene.hip, live_lens_samples); the volume-free
 * PIV frame of the reference's sample data (5e8 rays) 25.6 -> 15.5 ms, the image bit for bit.  rays_launched keeps counting
 * sources x rays_per_source.  photon_scene_live_rays: how many lens samples per source such a launch keeps (rays_per_source
 * when none can be ruled out: narrow cones, tilted or off-axis first element, BOS patterns generated on the device; PIV fields
 * generated on the device are bounded by the generator's box). */
int photon_scene_set_skip_doomed(photon_scene_t *scene, int on);
int photon_scene_live_rays(const photon_scene_t *scene);
/* the kept lens samples themselves, ascending (out: room for `capacity` >= photon_scene_live_rays entries); returns their number, -1 on a bad argument */
int photon_scene_live_samples(const photon_scene_t *scene, int *out, int capacity);
/* The same switch also leaves out, on the volume-free path, the SOURCES whose image cannot fall on the sensor (one biconvex
 * thick lens or one thin lens on the axis, no sensor-position noise, no dumps: photon_scene.hip, source_misses_sensor --
 * an interval bound on where the lens can put the source's rays; photon's sample PIV frame draws particles over a field 1.5 x
 * wider than the camera sees, run_simulation_02.py:956-958).  The image is unchanged.  The list is made once per scene, by its
 * first volume-free photon_trace (or by the query below): one small kernel and two small copies on the null stream, for which
 * that call waits -- a scene that only marches through a volume never pays them.  photon_scene_live_sources: the sources
 * that are launched, ascending (out may be NULL to ask for the count); -1 when nothing could be ruled out (all are), -2 on a
 * bad argument. */
long long photon_scene_live_sources(const photon_scene_t *scene, int *out, long long capacity);
/* The bound itself, host arithmetic only (no device call, usable without a GPU): off[i] = 1 when source (x, y, z)[i] cannot
 * reach a pixel through any of the lens samples (lens_x, lens_y)[k] on the plane z = image_distance.  Returns 0; 1 when the
 * geometry is not covered (off all zeros); 2 on a bad argument. */
int photon_sources_missing_sensor(const float *lens_x, const float *lens_y, int n_samples, float image_distance, float beam_wavelength,
                                  int num_elements, const element_data_t *element_data_p, const double (*element_center)[3],
                                  const double (*element_plane_parameters)[4], const int *element_system_index,
                                  const camera_design_t *camera_design_p, const float *x, const float *y, const float *z,
                                  long long n, unsigned char *off);

/* The launch loop (parallel_ray_tracing.cu:3515-3672) for sources [src_begin, src_end)
 * with everything resident in HBM.  d_image: device f32[H*W], accumulated into.
 * vol may be NULL (= simulate_density_gradients false).  stream: hipStream_t as void*
 * (NULL = default stream).  Asynchronous unless stats != NULL (stats forces a sync).
 * The traces of ONE scene share its ray-state workspace, work queues and f64 accumulator (which every trace leaves zeroed for
 * the next): issue them on one stream, or order them yourself when you change streams; different scenes are independent.
 * Hand-off errors of a segmented march (a wave gave up waiting for the previous piece of its group, or read a stale ray
 * state: never observed, and then the render is incomplete) are counted on the device and REPORTED where the host reads the
 * statistics -- photon_trace with stats, photon_scene_stats_end -- which zero the count when they start and fail (non-zero
 * return, message on stderr) when it is set.  A caller of plain asynchronous traces (stats = NULL, no window) learns of them
 * from photon_scene_check. */
int photon_trace(photon_scene_t *scene, const photon_volume_t *vol, int ray_tracing_algorithm,
                 int64_t src_begin, int64_t src_end, float *d_image, void *stream,
                 photon_trace_stats_t *stats);

/* Statistics over a WINDOW of photon_trace calls with no host synchronisation inside it (a timed loop): _begin zeroes
 * the counters on the stream; every photon_trace(..., stats = NULL) of this scene up to _end records its HIP events on
 * its stream and lets the counters run; _end waits for the stream and returns the SUMS over the window's traces
 * (march_ms, total_ms, the counters; shader_clock_mhz over all march waves; traces = number of calls).  The reference
 * prints one wall-clock time per call instead (parallel_ray_tracing.cu:3498-3503, 3678-3684). */
/* The window belongs to the stream it was opened on: a photon_trace of this scene on another stream, or more than 65536
 * traces in one window, is refused. */
int photon_scene_stats_begin(photon_scene_t *scene, void *stream);
int photon_scene_stats_end(photon_scene_t *scene, void *stream, photon_trace_stats_t *stats);
/* Waits for `stream` and returns 0 when no trace of this scene since the count was last zeroed (photon_trace with stats,
 * photon_scene_stats_begin, a previous photon_scene_check) had a hand-off error, 1 (and a message on stderr) otherwise;
 * zeroes the count. */
int photon_scene_check(photon_scene_t *scene, void *stream);

/* Wave timing of the march launches (measurement; off by default, costs a handful of atomics per wave when on).  With
 * it on, every march launch after the statistics counters were last zeroed (photon_trace with stats, or
 * photon_scene_stats_begin) records -- on the device's constant 100 MHz clock -- when its first wave entered, when each
 * wave started its first 64-ray group and when it left; photon_scene_march_profile returns the means over those launches
 * (the first 64 of them), all times counted from the first wave's entry:
 *   span_ms          until the last wave left (the launch as the chip saw it)
 *   start_mean/max   until a wave started its first group (dispatch ramp, argument loads, first queue access)
 *   end_min/mean     until a wave left; span_ms - end_mean_ms = the DRAIN, the average time a wave slot stood empty at the
 *                    end of the launch while the last groups finished
 * The caller sets struct_size = sizeof(photon_march_profile_t) (the library refuses a smaller struct than it knows). */
typedef struct photon_march_profile_t {
    uint32_t struct_size;
    uint32_t launches;              /* march launches the means cover (0: profile off, or no launch since the reset) */
    uint32_t waves;                 /* waves that served at least one group, mean per launch */
    float span_ms;
    float start_mean_ms, start_max_ms;
    float end_min_ms, end_mean_ms;
} photon_march_profile_t;
int photon_scene_set_march_profile(photon_scene_t *scene, int on);
int photon_scene_march_profile(photon_scene_t *scene, photon_march_profile_t *out);
/* The raw stamps of profiled launch `launch` (0 = the first since the reset): out[64][8] = per workgroup-index-mod-64 slot
 * {~min entry, ~min first-group start, sum of starts, max start, ~min exit, sum of exits, max exit, waves} in ticks of the
 * 100 MHz clock (slot & 7 = the XCD); for tools/tail_by_xcd.py. */
int photon_scene_march_profile_raw(photon_scene_t *scene, unsigned launch, unsigned long long *out);

/* March-only entry point for parity tests: n rays (host arrays pos/dir f32[n][3], world
 * frame) through trace_rays_through_density_gradients (.h:1455-1544); results in place,
 * steps (optional) receives the per-ray completed iteration count. */
int photon_trace_volume_rays(const photon_volume_t *vol, int ray_tracing_algorithm, int n,
                             float *pos, float *dir, int *steps);

/* The same march for arbitrary rays, but THROUGH the render path's march launch -- persistent waves over the work queues,
 * marches cut into `segments` pieces (-1 the library's choice, 1 whole, 2..64 forced): what the adversarial parity tests
 * drive (rays from every side, tiny grids, the below-minimum repair) to hold the segmented march to the oracle's bits.
 * ray_tracing_algorithm 1 or 2.  Results in place. */
int photon_trace_volume_rays_queued(const photon_volume_t *vol, int ray_tracing_algorithm, int n, float *pos, float *dir,
                                    int segments);

/* ------------------------------------------------------------------------------------
 * Section 3: scene generation on the device (the step right before the
 * hot path).  photon builds its source arrays and its synthetic density files in Python
 * (run_simulation_02.py, nrrd_functions.py) and ships them through start_ray_tracing; for
 * 1e6-source / 512^3 configurations these entry points build the same data directly in HBM.
 * ---------------------------------------------------------------------------------- */

typedef struct photon_sources photon_sources_t; /* light-field sources (x, y, z, radiance, diameter index) in HBM */

/* BOS target, generate_bos_lightfield_data (run_simulation_02.py:1328-1551): every dot centre
 * (dot_x[g], dot_y[g]) is expanded by the point template (tmpl_x[j], tmpl_y[j]) -- the sunflower
 * disc of calculate_sunflower_coordinates (:999-1056) -- into source g*n_tmpl + j at
 * (dot_x[g] + tmpl_x[j], dot_y[g] + tmpl_y[j], z); sums in double, stored as f32 (what the ctypes
 * marshalling does, perform_ray_tracing_03.py:1730-1745); radiance constant, diameter index 1. */
int photon_sources_bos(const double *dot_x, const double *dot_y, int n_dots, const double *tmpl_x,
                       const double *tmpl_y, int n_tmpl, double z, double radiance,
                       photon_sources_t **out);
/* PIV particle field, run_simulation_02.py:774-996: X, Y, Z uniform in [box_min, box_max),
 * radiance = irradiance_constant / (sigma sqrt(2 pi)) * exp(-Z^2 / (2 sigma^2)) with
 * sigma = beam_fwhm / (2 sqrt(2 ln 2)) (the laser sheet, :961-962), z = Z + z_object.  The reference
 * draws from numpy's unseeded generator; here particle i takes the four 32-bit words of
 * Philox4x32-10(seed, i) (include/photon_philox.h, stream PHOTON_STREAM_SCENE): reproducible, and any
 * particle can be regenerated on its own.  diameter_cdf (may be NULL): cumulative distribution over
 * n_diameters table columns, index = first d with u < cdf[d]; NULL = index 1 like the reference (:992). */
int photon_sources_piv(uint64_t seed, long long n, const double box_min[3], const double box_max[3],
                       double z_object, double beam_fwhm, double irradiance_constant,
                       const double *diameter_cdf, int n_diameters, photon_sources_t **out);
long long photon_sources_count(const photon_sources_t *sources);
/* Copy back to host arrays (any of them may be NULL). */
int photon_sources_download(const photon_sources_t *sources, float *x, float *y, float *z, double *radiance,
                            int *diameter_index);
void photon_sources_free(photon_sources_t *sources);

/* photon_scene_create with the sources taken from `sources` (device-to-device copy; the scene does not
 * keep a reference).  lightfield_source_p supplies the scalars only (z_offset, object_distance,
 * source_point_number): its arrays and num_particles are ignored. */
int photon_scene_create_from_sources(float lens_pitch, float image_distance,
                                     const scattering_data_t *scattering_data_p, const char *scattering_type_str,
                                     const lightfield_source_t *lightfield_source_p,
                                     const photon_sources_t *sources, int lightray_number_per_particle,
                                     float beam_wavelength, float aperture_f_number, int num_elements,
                                     const double (*element_center)[3], const element_data_t *element_data_p,
                                     const double (*element_plane_parameters)[4], const int *element_system_index,
                                     const camera_design_t *camera_design_p, float ray_cone_pitch_ratio,
                                     photon_scene_t **out);

/* Synthetic density field evaluated on the device instead of written to / read from an NRRD file
 * (nrrd_functions.py:14-57 is the writer it replaces): rho = rho0 + amp * exp(-|r - centre|^2 / (2 sigma^2))
 * on the grid origin + i * spacing, then the same volume build as photon_volume_load_nrrd.  origin is the
 * NRRD "space origin" (before the -750e3 z shift), centre in the same frame. */
int photon_volume_gaussian(int nx, int ny, int nz, const double spacing[3], const double origin[3],
                           double rho0, double amp, const double centre[3], double sigma,
                           int interpolation, photon_volume_t **out);

/* The same Gaussian density field written as an NRRD file (type float, dimension 3, raw, little endian, sizes /
 * spacings / space origin: what nrrd_functions.py:14-57 writes and loadNRRD reads), evaluated on the device. */
int photon_density_gaussian_write_nrrd(const char *path, int nx, int ny, int nz, const double spacing[3],
                                       const double origin[3], double rho0, double amp,
                                       const double centre[3], double sigma);

/* The library keeps freed scene-lifetime device blocks (ray-state workspace, source arrays, accumulators: what every
 * start_ray_tracing call allocates anew) in a cache and hands them to the next scene of the same shape, per device, up to
 * PHOTON_POOL_MAX_MB (default 4096; 0 = no cache): photon's unchanged Python pays ~1 ms of hipFree per call otherwise.
 * photon_trim_caches returns all cached blocks to the runtime (the library does so itself when one of ITS allocations finds
 * the device out of memory; another user of the device -- a framework's caching allocator -- should call it before a large
 * allocation of its own). */
void photon_trim_caches(void);

/* Library / build identification string (static storage): "photon-amd <version> (gfx950, HIP) <git commit>[-dirty] <flags>",
 * <flags> = "default" or "variant[...]" with the non-default -DPHOTON_* compile-time switches of this build. */
const char *photon_version(void);

/* Self-test hook: quot[i] = a[i] / b[i], rcp[i] = 1 / b[i], root[i] = sqrt(a[i]) (host arrays of n floats) evaluated with the
 * march loops' normal-range forms -- the compiler's correctly rounded division / square-root sequences without their range
 * scaling (photon_amd/csrc/device_vec.hpp) -- which return the IEEE results bit for bit whenever operands and results lie
 * within [2^-96, 2^96] in magnitude (a marching ray's sit within a few binades of 1); tests hold them against numpy. */
int photon_selftest_normal_range_math(int n, const float *a, const float *b, float *quot, float *rcp, float *root);

/* Self-test hook: perm_out[k] = index (in [first, first + n)) of the k-th of the sources first .. first + n - 1 of the host
 * arrays x, y (n_total entries) in the spatial order lens-major launches use -- Morton order on a 2^16 x 2^16 grid over the
 * range's bounding box, one scale for both axes, ties in the caller's order -- computed by the device path (bounding box,
 * keys, the library's own stable radix sort: photon_amd/csrc/photon_sort.hip); tests hold it against numpy's stable argsort. */
int photon_selftest_morton_order(const float *x, const float *y, long long n_total, long long first, long long n, int *perm_out);

/* Device-to-device float4 streaming copy of `bytes` bytes, `reps` times: read + write rate in GB/s -- the HBM rate a
 * trivial kernel reaches on this GPU, which bench.py prints next to the 8 TB/s specification. */
int photon_measure_copy_gbs(size_t bytes, int reps, double *gbs_out);

/* ------------------------------------------------------------------------------------
 * Section 4: sensor post-processing on the device (the step right after the
 * hot path).  Replaces perform_ray_tracing_03.py:2190-2259 (noise -> clip -> 10^(gain/20) ->
 * normalise to the brightest pixel -> round to pixel_bit_depth -> stretch to 16 bit -> uint16 ->
 * optional centre crop), in f32 in numpy's evaluation order, so that the raw image stays in HBM and
 * only the uint16 picture crosses the bus.
 *   d_image   device f32[height*width], row-major; rewritten only when image_noise > 0 (the reference adds the
 *             noise to I_raw itself, :2196-2206; here N(0, 100 image_noise) from Philox(noise_seed, pixel))
 *   crop_rows / crop_cols   0 = no crop; otherwise the reference's window rows [H/2 - r/2, H/2 + r/2 - 1) (integer
 *             division; one row / column fewer than asked, as its slice has it) -- *out_rows / *out_cols receive
 *             the size of the result (may be NULL)
 *   d_out     device uint16[out_rows*out_cols]
 * Synchronises `stream` before returning. */
int photon_postprocess_u16(float *d_image, int width, int height, float pixel_gain, int pixel_bit_depth,
                           int intensity_rescaling, float image_noise, uint64_t noise_seed, int crop_rows, int crop_cols,
                           uint16_t *d_out, int *out_rows, int *out_cols, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PHOTON_AMD_PARALLEL_RAY_TRACING_H_ */
