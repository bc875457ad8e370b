#!/bin/bash
set -e
python tools/ab_geometry.py cramped_room 65536 32,2,12,1 32,4,12,1,2 32,4,12,1,1 32,4,12,1,3 32,3,12,1,2 32,4,8,1,2 32,4,10,1,2 32,5,12,1,2 32,6,12,1,2
python tools/ab_geometry.py coordination_ring 65536 16,2,12 16,2,12,0,1 16,3,12,0,1 16,3,12,0,2 16,4,8,0,2 16,4,10,0,2
python tools/ab_geometry.py counter_circuit 65536 16,2,12 16,2,12,0,1 16,3,12,0,1
python tools/ab_geometry.py asymmetric_advantages 65536 16,2,12 16,3,12,0,1
