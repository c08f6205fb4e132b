// tools/bench_delaunay.cpp -- host Delaunay (planar_prior.cpp) on the vertex pattern of cfg 3: 3 pixels in every 5x5 cell of 1600x1200.
//   g++ -O2 -std=c++17 -fopenmp -I mp-mvs_amd/host -I include tools/bench_delaunay.cpp mp-mvs_amd/host/planar_prior.cpp -o build/bench_delaunay -lpthread
//   MPMVS_HOST_THREADS=8 build/bench_delaunay
#include "PatchMatch.h"
#include <chrono>
#include <cstdio>
#include <random>
using namespace mpmvs_host;
int main(){
  std::mt19937 rng(1); std::vector<Point> pts;
  for(int cy=0;cy<240;++cy)for(int cx=0;cx<320;++cx){int k[3];k[0]=rng()%25;do k[1]=rng()%25;while(k[1]==k[0]);do k[2]=rng()%25;while(k[2]==k[0]||k[2]==k[1]);
    for(int q:k)pts.push_back(Point(cx*5+q%5,cy*5+q/5));}
  for(int r=0;r<5;++r){auto t0=std::chrono::steady_clock::now();auto tr=Delaunay(Rect{0,0,1600,1200},pts);auto t1=std::chrono::steady_clock::now();
    printf("%zu pts %zu tris %.1f ms\n",pts.size(),tr.size(),std::chrono::duration<double,std::milli>(t1-t0).count());}
}
