// scene_io.h -- file formats either side of the hot path and the
// file-based ProcessProblem; same names as the reference (include/utility.h:59-62,
// include/PatchMatch.h:75-78) with cv::Mat replaced by Image.
#ifndef MPMVS_HOST_SCENE_IO_H_
#define MPMVS_HOST_SCENE_IO_H_

#include <string>
#include <vector>

#include "PatchMatch.h"

bool readDepthDmb(const std::string file_path, Image& depth);
int writeDepthDmb(const std::string file_path, const Image& depth);
bool readNormalDmb(const std::string file_path, Image& normal);
int writeNormalDmb(const std::string file_path, const Image& normal);
Camera ReadCamera(const std::string& cam_path);
// reference GenerateSampleList(const ConfigParams&, ...): the two config values it uses are passed directly
void GenerateSampleList(const std::string& input_folder, int maxSourceImageNum, int maxImageSize, std::vector<Scene>& Scenes);
// reference include/PatchMatch.h:29-33,77
struct PointList {
    float3 coord;
    float3 normal;
    float3 color;
};
void StoreColorPlyFileBinaryPointCloud(const std::string& plyFilePath, const std::vector<PointList>& pc);
// reference RunFusion(const ConfigParams&, const std::vector<Scene>&) (include/PatchMatch.h:85); returns the point count
long RunFusion(const std::string& input_folder, const std::string& output_folder, std::vector<Scene>& Scenes, bool use_dynamic_consistency, int device = 0);
bool readGrayImage(const std::string& path, Image& img);  // binary PGM (P5), 8 bit
void ProcessProblem(const std::string& input_folder, const std::string& output_folder, std::vector<Scene>& Scenes, const int ID,
                    bool geom_consistency, bool planar_prior, uint64_t seed = 0, int device = 0, int max_scale = 2);

#endif
