// scene_io.cpp -- the on-disk formats either side of the hot path:
//   DMB maps      reference src/utility.cpp:193-308   (int32 type=1, h, w, nb; h*w*nb fp32, row-major)
//   *_cam.txt     reference src/PatchMatch.cpp:109-143 (ReadCamera)
//   pair.txt      reference src/PatchMatch.cpp:67-107  (GenerateSampleList)
// and the file-based ProcessProblem of the reference (src/PatchMatch.cpp:506-638):
// reads images + cameras (+ the previous pass's depths/normals/costs.dmb for
// geometric consistency), runs the in-memory ProcessProblem, writes
// depths.dmb / normals.dmb / costs.dmb under <output>/2333_<id>.
// Images: the reference reads <input>/images/%08d.jpg through OpenCV; no JPEG
// decoder exists on the target image (row f-3), so this build reads the same
// name with extension .pgm (binary P5, 8 bit) -- `mogrify -format pgm *.jpg`.
#include <sys/stat.h>

#include <cfloat>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>

#include "PatchMatch.h"
#include "scene_io.h"

// ---------------------------------------------------------------------------
// DMB
// ---------------------------------------------------------------------------
static bool read_dmb(const std::string& path, Image& img, int channels) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        std::cout << "Error opening file " << path << std::endl;
        return false;
    }
    int32_t type = -1, h = 0, w = 0, nb = 0;
    bool ok = fread(&type, 4, 1, f) == 1 && fread(&h, 4, 1, f) == 1 && fread(&w, 4, 1, f) == 1 && fread(&nb, 4, 1, f) == 1;
    if (!ok || type != 1 || h <= 0 || w <= 0 || nb != channels) {
        fclose(f);
        return false;
    }
    img = Image(h, w, channels);
    ok = fread(img.data.data(), sizeof(float), (size_t)h * w * nb, f) == (size_t)h * w * nb;
    fclose(f);
    return ok;
}

static int write_dmb(const std::string& path, const Image& img) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) {
        std::cout << "Error opening file " << path << std::endl;
        return -1;
    }
    const int32_t hdr[4] = {1, img.rows, img.cols, img.ch};
    fwrite(hdr, 4, 4, f);
    fwrite(img.data.data(), sizeof(float), img.data.size(), f);
    fclose(f);
    return 0;
}

bool readDepthDmb(const std::string file_path, Image& depth) { return read_dmb(file_path, depth, 1); }
int writeDepthDmb(const std::string file_path, const Image& depth) { return write_dmb(file_path, depth); }
bool readNormalDmb(const std::string file_path, Image& normal) { return read_dmb(file_path, normal, 3); }
int writeNormalDmb(const std::string file_path, const Image& normal) { return write_dmb(file_path, normal); }

// ---------------------------------------------------------------------------
// cameras and the Problem list
// ---------------------------------------------------------------------------
Camera ReadCamera(const std::string& cam_path) {
    Camera camera{};
    std::ifstream file(cam_path);
    if (!file.is_open()) {
        std::cout << "can not open file in path:   " << cam_path << std::endl;
        exit(1);
    }
    std::string line;
    file >> line;  // "extrinsic"
    for (int i = 0; i < 3; ++i) file >> camera.R[3 * i + 0] >> camera.R[3 * i + 1] >> camera.R[3 * i + 2] >> camera.t[i];
    float tmp[4];
    file >> tmp[0] >> tmp[1] >> tmp[2] >> tmp[3];
    file >> line;  // "intrinsic"
    for (int i = 0; i < 3; ++i) file >> camera.K[3 * i + 0] >> camera.K[3 * i + 1] >> camera.K[3 * i + 2];
    // C = -R^T t (reference src/PatchMatch.cpp:134-136)
    camera.C[0] = -(camera.R[0] * camera.t[0] + camera.R[3] * camera.t[1] + camera.R[6] * camera.t[2]);
    camera.C[1] = -(camera.R[1] * camera.t[0] + camera.R[4] * camera.t[1] + camera.R[7] * camera.t[2]);
    camera.C[2] = -(camera.R[2] * camera.t[0] + camera.R[5] * camera.t[1] + camera.R[8] * camera.t[2]);
    float depth_num, interval;
    file >> camera.depth_min >> interval >> depth_num >> camera.depth_max;
    return camera;
}

void GenerateSampleList(const std::string& input_folder, int maxSourceImageNum, int maxImageSize, std::vector<Scene>& Scenes) {
    Scenes.clear();
    const std::string path = input_folder + "/pair.txt";
    std::ifstream file(path);
    if (!file.is_open()) {
        std::cout << "can not open file in path:   " << path << std::endl;
        exit(1);
    }
    int num_images;
    file >> num_images;
    for (int i = 0; i < num_images; ++i) {
        Scene scene;
        scene.max_image_size = maxImageSize;
        file >> scene.refID;
        scene.srcID.push_back(scene.refID);
        while (scene.refID > (int)Scenes.size()) {  // ids missing from pair.txt become placeholders
            Scene temp;
            temp.estimate = false;
            Scenes.push_back(temp);
        }
        int num_src_images;
        file >> num_src_images;
        for (int j = 0; j < num_src_images; ++j) {
            int id;
            float score;
            file >> id >> score;
            if (score <= 0.0f) continue;
            if (j < maxSourceImageNum) scene.srcID.push_back(id);
        }
        scene.estimate = num_src_images != 0;
        Scenes.push_back(scene);
    }
}

// ---------------------------------------------------------------------------
// images (binary PGM stands in for the reference's JPEG, see header comment)
// ---------------------------------------------------------------------------
bool readGrayImage(const std::string& path, Image& img) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0, 0, 0};
    int w = 0, h = 0, maxv = 0;
    auto next_int = [&](int& v) {
        int c = fgetc(f);
        while (c == '#' || c == ' ' || c == '\n' || c == '\r' || c == '\t') {
            if (c == '#')
                while (c != '\n' && c != EOF) c = fgetc(f);
            c = fgetc(f);
        }
        if (c < '0' || c > '9') return false;
        v = 0;
        while (c >= '0' && c <= '9') {
            v = v * 10 + (c - '0');
            c = fgetc(f);
        }
        return true;  // the single whitespace after the number has been consumed
    };
    bool ok = fread(magic, 1, 2, f) == 2 && magic[0] == 'P' && magic[1] == '5' && next_int(w) && next_int(h) && next_int(maxv) && maxv == 255 && w > 0 && h > 0;
    if (ok) {
        std::vector<unsigned char> buf((size_t)w * h);
        ok = fread(buf.data(), 1, buf.size(), f) == buf.size();
        if (ok) {
            img = Image(h, w, 1);
            for (size_t i = 0; i < buf.size(); ++i) img.data[i] = (float)buf[i];  // convertTo(CV_32FC1), reference :882
        }
    }
    fclose(f);
    return ok;
}

static std::string id8(int id) {
    std::stringstream s;
    s << std::setw(8) << std::setfill('0') << id;
    return s.str();
}

// reference src/PatchMatch.cpp:506-638 with its file traffic
void ProcessProblem(const std::string& input_folder, const std::string& output_folder, std::vector<Scene>& Scenes, const int ID,
                    bool geom_consistency, bool planar_prior, uint64_t seed, int device, int max_scale) {
    Scene& scene = Scenes[ID];
    std::cout << "Processing image " << id8(scene.refID) << " ..." << std::endl;
    const std::string result_folder = output_folder + "/2333_" + id8(scene.refID);
    mkdir(result_folder.c_str(), 0777);
    // PatchMatchInit's file half (reference :871-890, :934-950, :1052-1063)
    for (int sid : scene.srcID) {
        Scene& s = Scenes[sid];
        if (s.image.empty()) {
            if (!readGrayImage(input_folder + "/images/" + id8(sid) + ".pgm", s.image)) {
                std::cout << "Can not read this image !" << input_folder + "/images/" + id8(sid) + ".pgm" << std::endl;
                exit(EXIT_FAILURE);
            }
            s.cam = ReadCamera(input_folder + "/cams/" + id8(sid) + "_cam.txt");
        }
        if (geom_consistency) {
            const std::string prev = input_folder + "/MPMVS/2333_" + id8(sid);
            if (!readDepthDmb(prev + "/depths.dmb", s.depth)) {
                std::cout << "Can not read this depth image !" << std::endl;
                exit(EXIT_FAILURE);
            }
            if (sid == scene.refID) {
                if (!readNormalDmb(prev + "/normals.dmb", s.normal) || !readDepthDmb(prev + "/costs.dmb", s.cost)) {
                    std::cout << "Can not read this depth image !" << std::endl;
                    exit(EXIT_FAILURE);
                }
            }
        }
    }
    ProblemResult res;
    ProcessProblem(Scenes, ID, geom_consistency, planar_prior, seed, device, max_scale, &res);
    writeDepthDmb(result_folder + "/depths.dmb", res.depth);
    writeNormalDmb(result_folder + "/normals.dmb", res.normal);
    writeDepthDmb(result_folder + "/costs.dmb", res.cost);
    std::cout << "Processing image " << id8(scene.refID) << " done!" << std::endl;
}

// ---------------------------------------------------------------------------
// binary PLY of fused points (reference src/PatchMatch.cpp:145-198): x y z nx ny nz as
// float32, then red green blue as uchar; the colour triple is stored blue-first in
// PointList::color (OpenCV BGR) and written red-first; non-finite coordinates become 0
// ---------------------------------------------------------------------------
void StoreColorPlyFileBinaryPointCloud(const std::string& plyFilePath, const std::vector<PointList>& pc) {
    std::cout << "store 3D points to ply file" << std::endl;
    FILE* out = fopen(plyFilePath.c_str(), "wb");
    if (!out) {
        std::cout << "Error opening file " << plyFilePath << std::endl;
        return;
    }
    fprintf(out, "ply\nformat binary_little_endian 1.0\nelement vertex %zu\n", pc.size());
    fprintf(out, "property float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n");
    fprintf(out, "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n");
    for (const PointList& p : pc) {
        float3 X = p.coord;
        const bool finite = (X.x < FLT_MAX && X.x > -FLT_MAX) && (X.y < FLT_MAX && X.y > -FLT_MAX) && (X.z < FLT_MAX && X.z >= -FLT_MAX);
        if (!finite) X = float3{0.0f, 0.0f, 0.0f};
        const char rgb[3] = {(char)(int)p.color.z, (char)(int)p.color.y, (char)(int)p.color.x};
        fwrite(&X, sizeof(float), 3, out);
        fwrite(&p.normal, sizeof(float), 3, out);
        fwrite(rgb, 1, 3, out);
    }
    fclose(out);
}

// ---------------------------------------------------------------------------
// RunFusion over a dataset folder (reference src/PatchMatch.cpp:287-504): reads every
// estimated image's depths.dmb / normals.dmb, camera and image, fuses them on the GPU
// (mpmvs_fuse, the snapshot formulation of DESIGN.md section 8) and writes
// <output>/MPMVS_model.ply.  Colour = the grey value (this build reads grey PGM images).
// Returns the number of points, or -1.
// ---------------------------------------------------------------------------
long RunFusion(const std::string& input_folder, const std::string& output_folder, std::vector<Scene>& Scenes, bool use_dynamic_consistency, int device) {
    const int n = (int)Scenes.size();
    std::vector<Camera> cams(n);
    std::vector<int> estimate(n, 0), src_off(1, 0), src_ids;
    std::vector<Image> depths(n), normals(n), grays(n);
    for (int i = 0; i < n; ++i) {
        Scene& s = Scenes[i];
        if (s.estimate) {
            std::cout << "Reading image " << id8(i) << "..." << std::endl;
            const std::string res = input_folder + "/MPMVS/2333_" + id8(s.refID);
            if (!readDepthDmb(res + "/depths.dmb", depths[i]) || !readNormalDmb(res + "/normals.dmb", normals[i])) return -1;
            if (s.image.empty() && !readGrayImage(input_folder + "/images/" + id8(s.refID) + ".pgm", s.image)) return -1;
            cams[i] = ReadCamera(input_folder + "/cams/" + id8(s.refID) + "_cam.txt");
            grays[i] = s.image;
            if (grays[i].rows != depths[i].rows || grays[i].cols != depths[i].cols) {  // RescaleImageAndCamera, reference :262-284
                const float sx = depths[i].cols / (float)grays[i].cols, sy = depths[i].rows / (float)grays[i].rows;
                grays[i] = ResizeLinear(grays[i], depths[i].cols, depths[i].rows);
                cams[i].K[0] *= sx;
                cams[i].K[2] *= sx;
                cams[i].K[4] *= sy;
                cams[i].K[5] *= sy;
            }
            cams[i].width = depths[i].cols;
            cams[i].height = depths[i].rows;
            estimate[i] = 1;
            for (int id : s.srcID) src_ids.push_back(id);
        } else {  // placeholder image (reference :314-322): 1x1, never estimated, never a source
            cams[i] = Camera{};
            cams[i].width = cams[i].height = 1;
            depths[i] = Image(1, 1, 1);
            normals[i] = Image(1, 1, 3);
            grays[i] = Image(1, 1, 1);
            src_ids.push_back(i);
        }
        src_off.push_back((int)src_ids.size());
    }
    std::vector<const float*> dp(n), np_(n), gp(n);
    std::vector<std::vector<unsigned char>> valid(n), masks(n);
    std::vector<std::vector<float>> pts(n);
    std::vector<unsigned char*> vp(n), mp(n);
    std::vector<float*> pp(n);
    for (int i = 0; i < n; ++i) {
        const size_t wh = (size_t)cams[i].width * cams[i].height;
        dp[i] = depths[i].data.data();
        np_[i] = normals[i].data.data();
        gp[i] = grays[i].data.data();
        valid[i].assign(wh, 0);
        masks[i].assign(wh, 0);
        pts[i].assign(wh * 9, 0.0f);
        vp[i] = valid[i].data();
        mp[i] = masks[i].data();
        pp[i] = pts[i].data();
    }
    if (mpmvs_fuse(device, n, cams.data(), estimate.data(), dp.data(), np_.data(), gp.data(), src_off.data(), src_ids.data(), use_dynamic_consistency ? 1 : 0,
                   vp.data(), pp.data(), mp.data()) != 0)
        return -1;
    std::vector<PointList> cloud;
    for (int i = 0; i < n; ++i)
        for (size_t k = 0; k < valid[i].size(); ++k)
            if (valid[i][k]) {
                const float* p = &pts[i][k * 9];
                cloud.push_back(PointList{float3{p[0], p[1], p[2]}, float3{p[3], p[4], p[5]}, float3{p[6], p[7], p[8]}});
            }
    StoreColorPlyFileBinaryPointCloud(output_folder + "/MPMVS_model.ply", cloud);
    return (long)cloud.size();
}

// ---------------------------------------------------------------------------
// C entry points for the tests
// ---------------------------------------------------------------------------
extern "C" {
int mpmvs_host_write_dmb(const char* path, const float* data, int h, int w, int nb) {
    Image img(h, w, nb);
    std::memcpy(img.data.data(), data, img.data.size() * sizeof(float));
    return write_dmb(path, img);
}
// returns 0 and fills h, w, nb (and data when non-null and large enough), or -1
int mpmvs_host_read_dmb(const char* path, float* data, size_t capacity_floats, int* h, int* w, int* nb) {
    for (int ch = 1; ch <= 3; ch += 2) {
        Image img;
        FILE* f = fopen(path, "rb");
        if (!f) return -1;
        fclose(f);
        if (read_dmb(path, img, ch)) {
            *h = img.rows;
            *w = img.cols;
            *nb = img.ch;
            if (data && capacity_floats >= img.data.size()) std::memcpy(data, img.data.data(), img.data.size() * sizeof(float));
            return 0;
        }
    }
    return -1;
}
// points9: n rows of x y z nx ny nz c0 c1 c2 (colour in PointList order)
int mpmvs_host_write_ply(const char* path, const float* points9, int n) {
    std::vector<PointList> pc((size_t)n);
    for (int i = 0; i < n; ++i) {
        const float* p = points9 + 9 * (size_t)i;
        pc[i].coord = float3{p[0], p[1], p[2]};
        pc[i].normal = float3{p[3], p[4], p[5]};
        pc[i].color = float3{p[6], p[7], p[8]};
    }
    StoreColorPlyFileBinaryPointCloud(path, pc);
    return 0;
}
int mpmvs_host_read_camera(const char* path, mpmvs_camera* out) {
    *out = ReadCamera(path);
    return 0;
}
// pair.txt -> for each scene: estimate flag, refID, number of ids, ids (srcID[0] = refID); flat int array
int mpmvs_host_sample_list(const char* input_folder, int max_src, int max_size, int* out, int cap) {
    std::vector<Scene> S;
    GenerateSampleList(input_folder, max_src, max_size, S);
    int n = 0;
    auto put = [&](int v) {
        if (n < cap) out[n] = v;
        ++n;
    };
    put((int)S.size());
    for (const Scene& s : S) {
        put(s.estimate ? 1 : 0);
        put(s.refID);
        put((int)s.srcID.size());
        for (int id : s.srcID) put(id);
    }
    return n;
}
int mpmvs_host_read_pgm(const char* path, float* data, size_t capacity_floats, int* h, int* w) {
    Image img;
    if (!readGrayImage(path, img)) return -1;
    *h = img.rows;
    *w = img.cols;
    if (data && capacity_floats >= img.data.size()) std::memcpy(data, img.data.data(), img.data.size() * sizeof(float));
    return 0;
}
// the reference's main() pass loops (src/main.cpp:20-41) over a dataset folder, sequentially
// and in place (Gauss-Seidel through the files, as the reference does)
// reference main()'s last step (src/main.cpp:49): fuse the maps of a processed folder into
// <folder>/MPMVS/MPMVS_model.ply; returns the number of points or -1
long mpmvs_host_fuse_folder(const char* input_folder, int device, int max_src, int use_dynamic_consistency) {
    std::vector<Scene> Scenes;
    GenerateSampleList(input_folder, max_src, 3200, Scenes);
    const std::string in = input_folder;
    return RunFusion(in, in + "/MPMVS", Scenes, use_dynamic_consistency != 0, device);
}
int mpmvs_host_run_folder(const char* input_folder, int device, int max_src, int geom_iterations, int planar_prior,
                          int geomPlanarPrior, int max_scale, uint64_t seed) {
    std::vector<Scene> Scenes;
    GenerateSampleList(input_folder, max_src, 3200, Scenes);
    const std::string in = input_folder, out = in + "/MPMVS";
    mkdir(out.c_str(), 0777);
    bool pp = !geomPlanarPrior && planar_prior;
    for (size_t i = 0; i < Scenes.size(); ++i)
        if (Scenes[i].estimate) ProcessProblem(in, out, Scenes, (int)i, false, pp, seed + i, device, max_scale);
    for (int g = 0; g < geom_iterations; ++g) {
        pp = geomPlanarPrior && g != geom_iterations - 1;
        for (size_t i = 0; i < Scenes.size(); ++i)
            if (Scenes[i].estimate) ProcessProblem(in, out, Scenes, (int)i, true, pp, seed + 100003ull * (g + 1) + i, device, max_scale);
    }
    return 0;
}
}  // extern "C"
