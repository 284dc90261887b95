// mesh_raw.hpp — what read_mesh (io.rs:32-515) holds between reading and geometry: nodes and faces by number.
// Shared by the TGRID reader (mesh_io.cpp) and the in-memory form of the synthetic generators (mesh_gen.cpp): a generated mesh goes
// through the SAME geometry code as a file that was read — normals, areas, centroids, volumes, cell face lists in ascending face id —
// without the detour over a text file (BASELINE configs[4]: 580 MB per rank, 11 of 12.5 s of a rank's set-up in r04).
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

// Host image of mesh::Mesh as read_mesh builds it (the handle of orc_read_mesh / orc_mixed_channel_generate)
struct OrcMeshData {
    int32_t dimensions = 0;
    int64_t n_vertices = 0, n_faces = 0, n_cells = 0;
    std::vector<double> vertex;  // [3V]
    std::vector<int64_t> face_c0, face_c1, face_node_ptr, face_nodes, cell_face_ptr, cell_faces;
    std::vector<int32_t> face_zone;
    std::vector<double> face_area, face_normal, face_centroid, cell_centroid, cell_volume;
    struct Zone {
        uint64_t id;
        int32_t type;
        double scalar;
        double vec[3];
        std::string name;
    };
    std::vector<Zone> zones;                               // FaceZone, in order of first appearance in the file
    std::vector<std::pair<uint64_t, uint64_t>> cell_zones;  // (zone id, zone type), io.rs:180-193
};

namespace orc {

struct MeshV3 {
    double x, y, z;
};

struct RawFace {
    int64_t node_begin = -1;  // into the node pool
    int32_t n_nodes = 0;
    int32_t zone = -1;
    int64_t c[2] = {-1, -1};
};

struct RawMesh {
    int dims = 0;
    std::vector<MeshV3> vert;        // by node number - 1
    std::vector<char> vert_present;  // a node section may skip numbers: the geometry step insists on all of them
    std::vector<RawFace> faces;      // by face number - 1
    std::vector<int64_t> node_pool;  // the faces' node lists, 0-based node ids
    int64_t n_vert = 0, n_face = 0;
};

// io.rs:289-438: faces in ascending number (normal, centroid, area), cells from their faces (face lists, centroid, volume).
// `path` / `line_no` only name the source in an error text.  d.zones and d.cell_zones are the caller's.
int mesh_finalize_geometry(const char *path, int64_t line_no, RawMesh &R, OrcMeshData &d);

}  // namespace orc
