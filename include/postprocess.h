/*
 * postprocess.h -- drop-in for the reference's postprocess.h:
 *   writeOutputData(fileName, grid, h, N)  postprocess.h:5-47 -> mg3d_write_vtk (host; ASCII legacy VTK)
 */
#ifndef POSTPROCESS_H
#define POSTPROCESS_H

#include "mg3d.h"

static inline void writeOutputData(const char *fileName, const double *grid, const double h, const int N)
{
    (void)mg3d_write_vtk(fileName, grid, h, N);
}

#endif
