// ref_open_private.h -- TEST INFRASTRUCTURE ONLY (oracle/Makefile target `ref`).  Force-included AFTER the standard
// headers the reference's RANSAC sources use, so that only the reference's own class (RANSAC) loses its `private:`:
// ref_ransac.cpp calls RANSAC::solve_plane / solve_distance and reads plane_P / plane_N / plane_Q to print them.
#define private public
