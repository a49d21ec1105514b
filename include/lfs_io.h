/* lfs_io.h — C ABI of liblfs_io.so: the data formats either side of the training hot path.
 * Host-only C++17 (no GPU, no libtorch): the COLMAP sparse-model reader, the splat PLY writer / reader and image headers /
 * lossless decoders. Reference interfaces replaced (all libtorch / std::filesystem C++ there):
 *   read_colmap_cameras_and_images[_text]  src/loader/formats/colmap.cpp:913-957 (+ :305-455 binary, :459-640 text, :645-880 assembly)
 *   read_colmap_point_cloud[_text]         src/loader/formats/colmap.cpp:907-911, :933-937
 *   SplatData::save_ply / write_ply_impl   src/core/splat_data.cpp:113-169, attribute order :402-419
 *   load_ply                               src/loader/formats/ply.cpp:186-640
 *   get_image_info / load_image            src/core/image_io.cpp:59-72, :112-270 (OpenImageIO there)
 * Every function returns 0 on success, a negative LFS_IO_* code otherwise; lfs_io_last_error() holds the message of the
 * calling thread's last failure (the text the reference would have thrown as std::runtime_error).
 */
#ifndef LFS_IO_H
#define LFS_IO_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#define LFS_IO_API __attribute__((visibility("default")))

enum { LFS_IO_OK = 0, LFS_IO_E_INVALID = -1, LFS_IO_E_NOT_FOUND = -2, LFS_IO_E_FORMAT = -3, LFS_IO_E_UNSUPPORTED = -4 };

LFS_IO_API const char* lfs_io_last_error(void);
LFS_IO_API const char* lfs_io_version(void);

/* ---- COLMAP ------------------------------------------------------------------------------------------------------ */
/* CAMERA_MODEL ids of COLMAP (colmap.cpp:117-129) */
enum { LFS_COLMAP_SIMPLE_PINHOLE = 0, LFS_COLMAP_PINHOLE = 1, LFS_COLMAP_SIMPLE_RADIAL = 2, LFS_COLMAP_RADIAL = 3, LFS_COLMAP_OPENCV = 4,
       LFS_COLMAP_OPENCV_FISHEYE = 5, LFS_COLMAP_FULL_OPENCV = 6, LFS_COLMAP_FOV = 7, LFS_COLMAP_SIMPLE_RADIAL_FISHEYE = 8,
       LFS_COLMAP_RADIAL_FISHEYE = 9, LFS_COLMAP_THIN_PRISM_FISHEYE = 10 };

/* One training view: the CameraData the reference assembles per image (colmap.cpp:645-830). */
typedef struct lfs_colmap_view {
    uint32_t camera_id;          /* COLMAP camera id                                                    */
    int32_t colmap_model;        /* LFS_COLMAP_*                                                        */
    int32_t camera_model_type;   /* gsplat::CameraModelType: 0 PINHOLE, 1 ORTHO, 2 FISHEYE              */
    uint64_t width, height;      /* after the images_<k> scale factor and the image-size correction     */
    float focal_x, focal_y, center_x, center_y;
    float R[9];                  /* world-to-camera rotation, row-major (qvec2rotmat, colmap.cpp:29-50) */
    float T[3];                  /* world-to-camera translation                                         */
    int32_t n_radial;      float radial[6];
    int32_t n_tangential;  float tangential[2];
    int32_t n_params;      float params[12]; /* raw (scaled) COLMAP parameters                          */
} lfs_colmap_view;

typedef struct lfs_colmap_scene lfs_colmap_scene;
/* format: 0 = cameras.bin / images.bin, 1 = cameras.txt / images.txt. Files are searched case-insensitively in
 * base/sparse/0, base/sparse, base (filesystem_utils.hpp:26-69). images_folder "images_4" scales widths, heights and
 * intrinsics by 1/4 (colmap.cpp:265-283); if the first image exists its real size overrides the database's (:836-865). */
LFS_IO_API int lfs_colmap_open(const char* base, const char* images_folder, int format, lfs_colmap_scene** scene);
/* Blender / NeRF-synthetic transforms file (read_transforms_cameras_and_images, src/loader/formats/transforms.cpp:73-265): `path` is the json or a
 * directory holding transforms_train.json / transforms.json. Views come back through the same accessors (PINHOLE, camera ids 0..n-1, scene
 * centre = origin). */
LFS_IO_API int lfs_transforms_open(const char* path, lfs_colmap_scene** scene);
LFS_IO_API void lfs_colmap_close(lfs_colmap_scene* scene);
LFS_IO_API uint64_t lfs_colmap_num_views(const lfs_colmap_scene* scene);
LFS_IO_API int lfs_colmap_view_at(const lfs_colmap_scene* scene, uint64_t i, lfs_colmap_view* out);
LFS_IO_API const char* lfs_colmap_image_name(const lfs_colmap_scene* scene, uint64_t i);
LFS_IO_API const char* lfs_colmap_image_path(const lfs_colmap_scene* scene, uint64_t i);
LFS_IO_API int lfs_colmap_scene_center(const lfs_colmap_scene* scene, float center[3]); /* mean camera position (:875) */

typedef struct lfs_point_cloud lfs_point_cloud;
/* points3D.bin (format 0) / points3D.txt (format 1): positions f32 [N,3], colours u8 [N,3] */
LFS_IO_API int lfs_colmap_points_open(const char* base, int format, lfs_point_cloud** pc);
LFS_IO_API uint64_t lfs_point_cloud_size(const lfs_point_cloud* pc);
LFS_IO_API int lfs_point_cloud_copy(const lfs_point_cloud* pc, float* positions, uint8_t* colors);
LFS_IO_API void lfs_point_cloud_close(lfs_point_cloud* pc);

/* ---- splat PLY ----------------------------------------------------------------------------------------------------- */
/* binary_little_endian, one "vertex" element, float properties in the order x y z nx ny nz f_dc_* f_rest_* opacity scale_*
 * rot_* (splat_data.cpp:402-419). Row-major inputs: means [N,3], normals [N,3] or NULL (zeros), f_dc [N,n_dc],
 * f_rest [N,n_rest] (may be NULL when n_rest == 0), opacity [N], scaling [N,3], rotation [N,4]. */
LFS_IO_API int lfs_ply_write_splat(const char* path, uint64_t N, uint32_t n_dc, uint32_t n_rest, const float* means, const float* normals,
                                   const float* f_dc, const float* f_rest, const float* opacity, const float* scaling, const float* rotation);
typedef struct lfs_ply lfs_ply;
/* Generic reader of the "vertex" element (binary little endian or ascii; scalar properties of any PLY type, converted to f32). */
LFS_IO_API int lfs_ply_open(const char* path, lfs_ply** ply);
LFS_IO_API uint64_t lfs_ply_num_vertices(const lfs_ply* ply);
LFS_IO_API uint32_t lfs_ply_num_properties(const lfs_ply* ply);
LFS_IO_API const char* lfs_ply_property_name(const lfs_ply* ply, uint32_t i);
LFS_IO_API int lfs_ply_read(const lfs_ply* ply, float* out /* [N, num_properties] row-major */);
LFS_IO_API void lfs_ply_close(lfs_ply* ply);

/* ---- images -------------------------------------------------------------------------------------------------------- */
/* width / height / channels from the file header: PNG, JPEG, PNM (P5 / P6), BMP */
LFS_IO_API int lfs_image_info(const char* path, int32_t* width, int32_t* height, int32_t* channels);
/* load_image's output size (image_io.cpp:112-270): res_div in {<=1, 2, 4, 8}, then the max_width cap */
LFS_IO_API int lfs_image_target_size(int32_t width, int32_t height, int32_t res_div, int32_t max_width, int32_t* out_width, int32_t* out_height);
/* Decode to 8-bit RGB [h,w,3] (alpha dropped, 1 channel replicated, 2 channels -> (r, g, (r+g)/2) as image_io.cpp:222-247).
 * Native decoders: PNG (non-interlaced, 8/16 bit), binary PNM and JPEG (8-bit Huffman, baseline + progressive, 1 or 3 components,
 * sampling factors <= 2: the libjpeg default pipeline - islow IDCT, fancy upsampling, fixed-point YCbCr->RGB - bit-identical to
 * libjpeg-turbo); anything else returns LFS_IO_E_UNSUPPORTED (the Python host layer then tries Pillow). The buffer is owned by the
 * library: release with lfs_io_free. */
LFS_IO_API int lfs_image_load_rgb8(const char* path, uint8_t** data, int32_t* width, int32_t* height);
LFS_IO_API int lfs_image_write_png_rgb8(const char* path, const uint8_t* data, int32_t width, int32_t height);
LFS_IO_API void lfs_io_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
