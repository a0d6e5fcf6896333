/* design_host.h — host-side design helpers (design_host.cpp) used by the operators. */
#pragma once
#include <vector>
#include <array>

namespace saf {
void thin_svd(const float* M, int r, int c, std::vector<double>& U, std::vector<double>& S, std::vector<double>& V);
void pinv_f(const float* inM, int dim1, int dim2, float* outM);
bool sphere_triangulate(const std::vector<double>& P, std::vector<std::array<int, 3>>& faces);
bool find_ls_triplets(const float* ls_dirs_deg, int L, int omitLargeTriangles, std::vector<float>& verts, std::vector<int>& faces);
void invert_ls_mtx(const float* U, const int* grp, int nGroups, float* inv);
void vbap_gains(const float* src_dirs_deg, int S, int L, const int* grp, int nFaces, float spread, const float* inv, float* G);
bool vbap_table(const float* src_dirs_deg, int S, const float* ls_dirs_deg, int L, int omitLarge, int enableDummies, float spread,
                std::vector<float>& gtable, int* nTriangles);
void vbap_grid_dirs(int az_res_deg, int el_res_deg, std::vector<float>& src);
void maxre_weights(int order, std::vector<float>& a);
void decoder_matrix(const float* ls_dirs_deg, int nLS, int method, int order, int maxrE, float* dec);
void sh_eval_host(int kind, int order, const float* dirs, int nDirs, float* Y);
void sh_eval_dev(int kind, int order, const float* d_dirs, int nDirs, float* d_Y);
void launch_enc_update_Y(const int* d_order /*[nInst]*/, const float* d_dirs /*[nInst][64][2]*/, const int* d_recalc /*[nInst][64]*/, float* d_Y, long long y_inst, int nInst);
}  // namespace saf
