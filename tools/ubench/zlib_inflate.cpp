// host micro-benchmark: zlib's inflate forms against gtars_amd/csrc/inflate_fast.h on a synthetic fragment file
// g++ -O3 -o /tmp/zi tools/ubench/zlib_inflate.cpp -lz && /tmp/zi
#include <zlib.h>
#include "../../gtars_amd/csrc/inflate_fast.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
static double now(){return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();}
int main(){
  printf("zlib %s\n", zlibVersion());
  std::string text; unsigned s=12345; auto rnd=[&]{s=s*1664525u+1013904223u; return s>>8;};
  long pos=10000;
  for(int i=0;i<100000;i++){ char b[128]; pos+=rnd()%3000; int n=snprintf(b,sizeof b,"chr%d\t%ld\t%ld\tBC%04u-1\t%d\n",1+i/5000,pos,pos+100+rnd()%500,rnd()%500,1+rnd()%3); text.append(b,n);}  
  printf("text %zu bytes\n", text.size());
  gzFile f=gzopen("t.gz", getenv("GZLEVEL") ? getenv("GZLEVEL") : "wb6"); gzwrite(f,text.data(),text.size()); gzclose(f);
  FILE*fp=fopen("t.gz","rb"); std::vector<unsigned char> comp(8<<20); size_t cn=fread(comp.data(),1,comp.size(),fp); fclose(fp); printf("gz %zu bytes\n",cn);
  std::vector<char> out(text.size()+1024);
  for(int rep=0;rep<3;rep++){
    double t=now(); gzFile g=gzopen("t.gz","rb"); gzbuffer(g,1<<20); size_t tot=0; std::string o; std::vector<char> buf(1<<20); for(;;){int n=gzread(g,buf.data(),buf.size()); if(n<=0)break; o.append(buf.data(),n);} gzclose(g); double t1=now();
    z_stream z{}; inflateInit2(&z,16+MAX_WBITS); z.next_in=comp.data(); z.avail_in=cn; z.next_out=(Bytef*)out.data(); z.avail_out=out.size(); int r=inflate(&z,Z_FINISH); inflateEnd(&z); double t2=now();
    z_stream y{}; inflateInit2(&y,-MAX_WBITS); y.next_in=comp.data()+10; y.avail_in=cn-10-8; y.next_out=(Bytef*)out.data(); y.avail_out=out.size(); int r2=inflate(&y,Z_FINISH); size_t rawn=y.total_out; inflateEnd(&y); double t3=now();
    unsigned long c=crc32(0,(const Bytef*)out.data(),rawn); double t4=now();
    {
      comp.resize(comp.size());  // (padded: 8 MB buffer)
      std::string o2; size_t done=0, used=0; o2.resize(text.size()+512);
      double t5=now(); bool ok=gtars::fastinf::inflate_raw(comp.data()+10, cn-10, &used, o2, done); double t6=now();
      printf("fast raw inflate %.2f ms (ok=%d, %zu bytes, same=%d, used %zu of %zu) | ", (t6-t5)*1e3, (int)ok, done, (int)(done==text.size() && !memcmp(o2.data(), text.data(), done)), used, cn-18);
    }
    printf("gzread+append %.2f ms | inflate(gzip) %.2f ms (r=%d) | raw inflate %.2f ms (r=%d, %zu) | crc32 alone %.2f ms (%lx)\n",(t1-t)*1e3,(t2-t1)*1e3,r,(t3-t2)*1e3,r2,rawn,(t4-t3)*1e3,c);
  }
}
