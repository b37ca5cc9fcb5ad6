#!/bin/bash
# Records whether the pieces needed to time the REFERENCE itself on a software rasterizer exist on
# this machine (cargo + a Vulkan software ICD + a display server).  They do not in this
# image; the CPU baseline bench.py reports is therefore the oracle (kind "port").
for t in cargo rustc vulkaninfo Xvfb weston; do
    if command -v "$t" > /dev/null 2>&1; then echo "$t: $(command -v $t)"; else echo "$t: absent"; fi
done
ls /usr/share/vulkan/icd.d /etc/vulkan/icd.d 2> /dev/null | sed 's/^/vulkan icd: /' || true
[ -d /root/reference ] && echo "reference sources: present" || echo "reference sources: absent"
