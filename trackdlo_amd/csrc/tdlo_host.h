// tdlo_host.h -- host-side helpers of the path (see tdlo_host.cpp).
#pragma once
#include <vector>

namespace tdlo {

struct Vec3 { double x, y, z; };

// trackdlo::calc_LLE_weights (trackdlo.cpp:119-159); Y is M x 3 column-major, L is M x M column-major.
void lle_weights(int k, const double *Y, int M, double *L);
// H = (I - L)^T (I - L) (trackdlo.cpp:237)
void lle_regulariser(const double *L, int M, double *H);
// the same H for k = 6 as its 13 diagonals, Hb[13 i + u] = H(i, i - 6 + u) (bit for bit the dense routines' values), in O(M)
void lle_regulariser_band(const double *Y, int M, double *Hb);
// line_sphere_intersection (utils.cpp:185-241)
int line_sphere(const Vec3 &A, const Vec3 &B, const Vec3 &C, double radius, Vec3 out[2]);
// trackdlo::traverse_euclidean (trackdlo.cpp:584-898); out receives rows [idx, x, y, z].
// Returns the number of rows or -1 where the reference would index out of bounds.
int traverse_euclidean(const std::vector<double> &coord, const double *guide, int Mg,
                       const std::vector<int> &vis, int alignment, int anchor, std::vector<double> &out);

// The callback's self-occlusion test (trackdlo_node.cpp:279-343): indices of the nodes that are within visibility_threshold of the cloud AND whose
// projected pixel is not under an edge nearer the camera (edges = thick lines of dlo_pixel_width pixels).  proj: 3 x 4 row-major.  Ascending.
void self_occlusion_visible(const double *Y, int M, const double proj[12], int dlo_pixel_width, const double *node_dist, double visibility_threshold, std::vector<int> &vis);

// evaluator::get_piecewise_error (evaluator.cpp:258-283); chains are n x 3 column-major.
double piecewise_error(const double *Ytrack, int n1, const double *Ytrue, int n2);

}  // namespace tdlo
