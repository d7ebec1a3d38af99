/* Driver for the sanitizer build of the oracle (tests/test_oracle_golden.py): every orc_*_mt entry at 1920x1080 with the thread
 * counts the GPU tests use (0 = all cores) and odd ones, then ragged small sizes.  Round 5: the last native call before round 4's
 * unexplained abort was orc_warp_blend_mt(.., threads=0) at this size. */
#include "nus_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(void){
  const uint32_t w=1920,h=1080;
  uint8_t *a=malloc((size_t)w*h*4),*b=malloc((size_t)w*h*4),*o=malloc((size_t)w*h*4),*o1=malloc((size_t)w*h*4);
  uint8_t *big=malloc((size_t)w*h*16),*big1=malloc((size_t)w*h*16);
  float *flow=malloc((size_t)w*h*8);
  orc_gen_noise(a,w,h,1); orc_gen_gradient(b,w,h,3);
  for(size_t i=0;i<(size_t)w*h*2;++i) flow[i]=(float)((int)(i*2654435761u>>20)%4000-2000)/100.0f;
  int ths[]={0,1,3,7,16,17};
  orc_warp_blend(a,b,NULL,w,h,0.5f,o1);
  for(unsigned i=0;i<sizeof ths/sizeof*ths;++i){
    int t=ths[i];
    orc_warp_blend_mt(a,b,NULL,w,h,0.5f,o,t); if(memcmp(o,o1,(size_t)w*h*4)) {puts("warp diff");return 1;}
    orc_warp_blend_mt(a,b,flow,w,h,0.3f,o,t);
    orc_nearest_mt(a,w,h,big,2*w,2*h,t);
    orc_bilinear_mt(a,w,h,big,2*w,2*h,t);
    if(orc_lanczos3_mt(a,w,h,big,2*w,2*h,t)) return 2;
    for(int f=0;f<3;++f) if(orc_resize_mt(a,w,h,big,2*w,2*h,f,t)) return 3;
    if(orc_resize_mt(a,w,h,big,w*3/2,h*3/2,0,t)) return 3;
    if(orc_resize_mt(a,w,h,big,w/2,h/2,0,t)) return 3;
    if(orc_resize_mt(a,w,h,big,w/3+1,h/5+1,1,t)) return 3;
    printf("threads %d ok (max %d)\n",t,orc_max_threads());fflush(stdout);
  }
  /* small / ragged */
  for(uint32_t ww=1;ww<40;ww+=3) for(uint32_t hh=1;hh<20;hh+=2){
    uint8_t *s=malloc((size_t)ww*hh*4),*d=malloc((size_t)ww*hh*16*4);
    orc_gen_noise(s,ww,hh,ww*hh);
    for(int t=0;t<9;t+=2){ orc_resize_mt(s,ww,hh,d,2*ww,2*hh,0,t); orc_warp_blend_mt(s,s,NULL,ww,hh,0.5f,d,t); orc_nearest_mt(s,ww,hh,d,4*ww,4*hh,t); orc_bilinear_mt(s,ww,hh,d,3*ww,hh,t);
      orc_fsr1(s,ww,hh,d,2*ww,2*hh,0.f,0.7f);}
    free(s);free(d);
  }
  free(a);free(b);free(o);free(o1);free(big);free(big1);free(flow);
  puts("all ok"); return 0;
}
